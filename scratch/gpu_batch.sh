#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "properties_full_size" > $O/r06_dbg.log 2>&1
grep -n "^E \|passed\|failed" $O/r06_dbg.log | head -20
timeout -k 5 1500 python -X faulthandler -m pytest tests -m gpu -q > $O/r06_gpu_tests_fused_final.log 2>&1
grep -n "passed\|failed\|FAILED\|Fatal\|Aborted\|core\|File \"/root" $O/r06_gpu_tests_fused_final.log | head -40
