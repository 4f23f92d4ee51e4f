#!/bin/bash
# round 6, call 33: the full chains (face 1000-step DDPM, body ddim100) against the committed oracle states: tool (three precisions, every saved step) + the new GPU tests
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 300 python tests/tools/chain_vs_oracle.py --side gpu --workload body 2>&1 | grep "^gpu"
timeout 300 python tests/tools/chain_vs_oracle.py --side gpu --workload face 2>&1 | grep "^gpu"
timeout -k 5 600 python -m pytest tests/test_hip_round6.py -m gpu -q -k "full_sampling_chain" 2>&1 | tail -4
