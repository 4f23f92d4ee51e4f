#!/bin/bash
# build scratch/attn3_bench with temps under /tmp/a3 and print the register / block census of attn3_kernel<64>
set -e
mkdir -p /tmp/a3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DA2P_HALF $EXTRA /root/repo/scratch/attn3_bench.hip -o /tmp/a3/attn3_bench -save-temps=obj -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|warning: inline" | head -20
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DA2P_HALF $EXTRA /root/repo/scratch/attn3_bench.hip -o /tmp/a3/attn3_bench -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A10 "Function Name: _Z12attn3" | grep -E "Function Name|VGPRs:|AGPRs|Spill|ScratchSize" | sed 's/.*remark: [^ ]* *//'
cp /tmp/a3/attn3_bench /root/repo/scratch/attn3_bench
F=/tmp/a3/attn3_bench-hip-amdgcn-amd-amdhsa-gfx950.s
A=$(grep -n "^_Z12attn3_kernelILi64" $F | head -1 | cut -d: -f1); E=$(awk -v a=$A 'NR>a && /^\.Lfunc_end/{print NR; exit}' $F)
python3 /root/repo/scratch/isa_blocks.py $F $A $E
