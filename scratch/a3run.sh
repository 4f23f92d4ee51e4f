#!/bin/bash
# GPU side: correctness of the plain-step build on all shapes, then the ablation variants on the headline shape
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out; mkdir -p $O
echo "== plain step (compiler-scheduled) ==" ; timeout 120 scratch/a3v/attn3_plain | head -8
for v in 0 1 2 4 6 8 14 16 30; do echo "== A3X=$v =="; timeout 60 scratch/a3v/attn3_x$v p; done
