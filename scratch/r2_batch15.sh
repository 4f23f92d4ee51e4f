R=$GRAFT_REPO_ROOT; cd $R
j='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], {k:(v["launches_per_step"], v["avg_launch_us"]) for k,v in d["kernels"].items()})'
for prec in fp16 fp32; do
timeout 300 python bench.py --batch 1 --frames 240 --steps 50 --precision $prec --no-cpu-baseline --no-legs --no-parity 2>/dev/null | python -c "$j" "cfg0 $prec"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg0 -o p -- python $R/bench.py --batch 1 --frames 240 --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-legs --no-parity --no-kernel-timing > $R/gpurun_out/prof_cfg0.log 2>&1
cd $R
head -25 gpurun_out/prof_cfg0/p_kernel_stats.csv | cut -c1-150
python scratch/trace_gaps.py gpurun_out/prof_cfg0/p_kernel_trace.csv 2>&1 | tail -15
