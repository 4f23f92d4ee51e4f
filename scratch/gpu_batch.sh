#!/bin/bash
# round 6, call 19: per-workgroup timeline of attn3_kernel (scratch/a3v/attn3_wg: -DA3_WGSTAMPS) + the GPU side of tests/tools/chain_vs_oracle.py
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 120 scratch/a3v/attn3_wg wg 2>&1 | tee $O/r06_attn3_wg_timeline_v4.txt
timeout 300 python tests/tools/chain_vs_oracle.py --side gpu 2>&1 | tail -5
