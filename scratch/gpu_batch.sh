#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_hip_round5.py -m gpu -q > $O/r05_c22_tests.log 2>&1; grep -E "passed|failed|FAILED|AssertionError: " $O/r05_c22_tests.log | head -30
