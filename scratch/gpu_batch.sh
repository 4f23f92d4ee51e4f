#!/bin/bash
# round 4, call 18: key-split attention (KS = 2) against the plain form, kernel level
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 5 300 ./scratch/attn_occ ks 2>&1 | tee gpurun_out/c18_attn_ks.txt
