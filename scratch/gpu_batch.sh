#!/bin/bash
# round 6, call 72: 4-deep GEMM ring for the input projection (A2P_GEMM_RING4_BLOCKS=512 vs the default 256), same box; the two re-written parity tests
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 5 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "properties_full_size" 2>&1 | tail -2
for b in 8 32; do for v in 512 256 512 256; do
  export A2P_GEMM_RING4_BLOCKS=$v
  timeout -k 5 300 python bench.py --batch $b --no-cpu-baseline --no-parity --no-legs --steps 100 --warmup 8 > $O/r06_c72.json 2>/dev/null
  python - <<PY
import json
j=json.loads([l for l in open("$O/r06_c72.json") if l.startswith("{")][-1])
k=j["kernels"]
print("B=$b ring4_blocks=$v", j["value"], "steps/s", {x:(k[x]["avg_launch_us"],k[x]["launches_per_step"]) for x in ("gemm",)})
PY
done; done | tee $O/r06_gemm_ring4_ab.txt
