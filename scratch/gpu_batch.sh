#!/bin/bash
# round 6, call 22: phase stamps + EFFECTIVE shader clock of the tall chain launches (B=8: layer-1 POST = launch 4, layer-0 MID = launch 2; B=32 POST) inside a running step
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
export A2P_LIB_F16=$R/scratch/ab/liba2p_stamps_f16.so
( PP_BATCH=8 A2P_STAMP_LAUNCH=4 timeout 200 python scratch/phase_probe4.py 2>&1 | grep "^B="
  PP_BATCH=8 A2P_STAMP_LAUNCH=2 timeout 200 python scratch/phase_probe4.py 2>&1 | grep "^B="
  PP_BATCH=32 A2P_STAMP_LAUNCH=4 timeout 200 python scratch/phase_probe4.py 2>&1 | grep "^B=" ) | tee $O/r06_chain_phase_stamps_clock.txt
