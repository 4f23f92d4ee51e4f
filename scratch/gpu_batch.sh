#!/bin/bash
# round 5, call 4: attn2_kernel v2 (anti-phase matrix / vector segments) vs attn_kernel, standalone
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 5 300 scratch/attn2_bench abl > $O/r05_c8_attn2_abl.txt 2>&1; cat $O/r05_c8_attn2_abl.txt
