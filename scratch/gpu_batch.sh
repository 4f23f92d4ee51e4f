#!/bin/bash
# round 6, call 83: phase stamps of the final kernels: layer-1 MID (A2P_STAMP_LAUNCH=5?) and layer-1 POST at B=8 / B=32
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
export A2P_LIB_F16=$R/scratch/ab/liba2p_stamps_f16.so
for sel in 1 2 3 4 5 6; do for b in 8; do echo "A2P_STAMP_LAUNCH=$sel"; A2P_STAMP_LAUNCH=$sel PP_BATCH=$b timeout -k 5 300 python scratch/phase_probe4.py 2>/dev/null | grep "gen 4 block 0"; done; done | cut -c1-420 | tee $O/r06_final_phase_stamps.txt
