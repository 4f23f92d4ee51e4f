#!/bin/bash
# round 4, call 20: pipeline with the deferred check; deferred-max attention A/B (kernel level, alternating)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 5 600 python bench.py --pipeline --batch 8 > gpurun_out/c20_pipeline.json 2> gpurun_out/c20_pipeline.err; tail -c 700 gpurun_out/c20_pipeline.json; tail -3 gpurun_out/c20_pipeline.err
timeout -k 5 300 python -m pytest tests/test_hip_round4.py -x -q -m gpu -k "non_finite or two_stream" 2>&1 | tail -3
for i in 1 2; do
echo "--- plain"; timeout -k 5 120 ./scratch/attn_occ 2>&1 | grep -E 'dh=|T=  600|T=  768|T= 1536'
echo "--- deferred max"; timeout -k 5 120 ./scratch/attn_occ_defer 2>&1 | grep -E 'dh=|T=  600|T=  768|T= 1536'
done > gpurun_out/c20_defer_ab.txt 2>&1
cat gpurun_out/c20_defer_ab.txt
