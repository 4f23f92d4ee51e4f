R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b32 -o p -- python $R/bench.py --batch 32 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-legs --no-parity --repeats 1 > $R/gpurun_out/prof_b32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_body -o p -- python $R/bench.py --model pose --batch 16 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-legs --no-parity --repeats 1 > $R/gpurun_out/prof_body.log 2>&1
cd $R
rm -f gpurun_out/prof_b32/*kernel_trace.csv gpurun_out/prof_body/*kernel_trace.csv
head -8 gpurun_out/prof_b32/p_kernel_stats.csv | cut -c1-130
head -10 gpurun_out/prof_body/p_kernel_stats.csv | cut -c1-130
