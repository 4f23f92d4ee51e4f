#!/bin/bash
# round 4, call 3: the whole GPU suite after the clean-up (generation 2/3 out, non-finite flag, logit maximum, deterministic chain
# shape, mean_tokens) + the round-4 tests (strong scaling, pipeline job placements, trained-like statistics)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
T=c3
timeout -k 5 1500 python -m pytest tests/test_hip_round4.py -m gpu -q -x 2>&1 | tail -25 | tee gpurun_out/${T}_tests_r4.log
timeout -k 5 900 python -m pytest tests -m gpu -q -x --deselect tests/test_hip_round4.py 2>&1 | tail -8 | tee gpurun_out/${T}_tests_all.log
cp gpurun_out/parity_tests.json gpurun_out/${T}_parity_tests.json 2>/dev/null
