R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_v3 -o v3 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_v3.log 2>&1
cd $R
python scratch/trace_gaps.py gpurun_out/prof_v3/v3_kernel_trace.csv
rm -f gpurun_out/prof_v3/v3_kernel_trace.csv
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"
