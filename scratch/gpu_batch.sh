#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_round2.py tests/test_hip_round3.py -m gpu -q -k "two_ranks or bench_two" > gpurun_out/b10_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/b10_tests.log
