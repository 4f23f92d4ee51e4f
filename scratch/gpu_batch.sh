#!/bin/bash
# round 4, call 31: GPU suite after the fused keyframe attention
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 5 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/c31_tests_all.log
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee -a gpurun_out/c31_tests_all.log
cp gpurun_out/parity_tests.json gpurun_out/c31_parity_tests.json 2>/dev/null
