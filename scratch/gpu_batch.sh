#!/bin/bash
# round 4, call 21: packed vs single-lane fp32 VALU forms in the attention softmax (kernel level, alternating)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for i in 1 2; do
echo "--- packed (shipped)"; timeout -k 5 120 ./scratch/attn_occ 2>&1 | grep -E 'dh=|T=  600|T=  768|T= 1536'
echo "--- single-lane forms"; timeout -k 5 120 ./scratch/attn_occ_nopk 2>&1 | grep -E 'dh=|T=  600|T=  768|T= 1536'
done > gpurun_out/c21_nopk_ab.txt 2>&1
cat gpurun_out/c21_nopk_ab.txt
