#!/bin/bash
# round 6, call 31: attn3_kernel v5 with the tile requests spread over the step (A3_SPREAD_DMA=1: wave w issues its four pieces in query tile w + 1's group) vs v5
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 120 scratch/a3v/attn3_spread 2>&1 | tee $O/r06_attn3_bench_v5_spread.txt
for v in wg wg_spread wg wg_spread; do echo "== $v"; timeout 120 scratch/a3v/attn3_$v wg 2>&1 | grep "^B="; done | tee $O/r06_attn3_wg_timeline_v5_spread.txt
