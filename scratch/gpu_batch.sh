#!/bin/bash
# round 5, call 32: phase stamps of the final tall POST kernels incl. the feed-forward block per hidden chunk (linear1 | GELU + barriers | linear2)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
export A2P_LIB_F16=$R/scratch/ab/liba2p_stamps_f16.so
for b in 8 32; do PP_BATCH=$b A2P_STAMP_LAUNCH=4 timeout -k 5 300 python scratch/phase_probe4.py 2>&1 | grep "gen 4"; done > $O/r05_c32_phase_probe4.txt
cat $O/r05_c32_phase_probe4.txt
