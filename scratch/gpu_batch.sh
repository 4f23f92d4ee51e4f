#!/bin/bash
# round 4, call 32: final evidence after the fused keyframe attention: default bench line, body kernel stats, pipeline
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-legs --no-parity --repeats 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_body -o p -- $B --model pose --batch 16 > $O/prof_body.log 2>&1
cp $O/prof_body/p_kernel_stats.csv $O/r04_body_kernel_stats.csv; rm -rf $O/prof_body
cd $R
head -7 $O/r04_body_kernel_stats.csv | cut -c1-120
timeout -k 5 900 python bench.py > $O/r04_bench_default.json 2> $O/r04_bench_default.err; tail -c 200 $O/r04_bench_default.json
timeout -k 5 600 python bench.py --pipeline --batch 8 > $O/r04_pipeline.json 2> $O/r04_pipeline.err; tail -c 500 $O/r04_pipeline.json
