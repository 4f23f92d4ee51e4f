#!/bin/bash
# round 4, call 27: lip regressor over all whole chunks in one batch: front-end tests, then the pipeline line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 5 900 python -m pytest tests/test_frontend_hip.py -x -q -m gpu 2>&1 | tail -4
timeout -k 5 600 python bench.py --pipeline --batch 8 > gpurun_out/c27_pipeline.json 2> gpurun_out/c27_pipeline.err; tail -c 600 gpurun_out/c27_pipeline.json; tail -2 gpurun_out/c27_pipeline.err
