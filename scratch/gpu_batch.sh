#!/bin/bash
# round 5, call 17: full GPU suite + default bench with the tall chain kernels as the default
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
rm -f $O/parity_tests.json
timeout -k 5 1500 python -m pytest tests -m gpu -q > $O/r05_c17_tests.log 2>&1; tail -5 $O/r05_c17_tests.log
timeout -k 5 900 python bench.py > $O/r05_c17_bench.json 2> $O/r05_c17_bench.err; tail -c 300 $O/r05_c17_bench.json
