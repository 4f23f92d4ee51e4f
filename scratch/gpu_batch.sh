#!/bin/bash
# The command list of the current gpurun call (one evolving script; git history keeps the earlier lists).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_tests.json
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/b7_tests.log 2>&1; echo "tests rc=$?"
tail -12 gpurun_out/b7_tests.log
