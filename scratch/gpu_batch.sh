#!/bin/bash
# round 6, call 73: the last layer's POST kernel is tall + fused whatever the calibrated family: test, and the step under the forced mixed family (what a slow-type GPU runs) with / without
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_hip_round6.py tests/test_hip_round5.py -m gpu -q -x -k "mixed_family or final_layer_inside or tall_chain" 2>&1 | tail -2
for b in 8 32; do for v in "41 0" "41 1" "4 0" "41 0" "41 1" "4 0"; do
  set -- $v
  export A2P_CHAIN_V=$1
  if [ $2 = 1 ]; then export A2P_NO_FUSED_FINAL=1; else unset A2P_NO_FUSED_FINAL; fi
  timeout -k 5 300 python bench.py --batch $b --no-cpu-baseline --no-parity --no-legs --steps 60 --warmup 8 > $O/r06_c73.json 2>/dev/null
  python - <<PY
import json
j=json.loads([l for l in open("$O/r06_c73.json") if l.startswith("{")][-1])
k=j["kernels"]; sub=k["_sub_classes"]
print("B=$b A2P_CHAIN_V=$1 no_fused_final=$2", j["value"], "steps/s", {a:v["avg_launch_us"] for a,v in sub.items()}, {x:(k[x]["avg_launch_us"],k[x]["launches_per_step"]) for x in ("gemm",)}, "family", j["roofline"].get("chain_family"))
PY
done; done | tee $O/r06_mixed_family_last_tall.txt
