#!/bin/bash
# round 4, call 24: 64 queries per wave (2 waves per workgroup, 2 per SIMD) against the shipped 32 (4 per workgroup, 3 per SIMD)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 5 300 ./scratch/attn_occ qt 2>&1 | tee gpurun_out/c24_attn_qt4.txt
