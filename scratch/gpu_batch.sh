#!/bin/bash
# round 6, call 71: fused final_layer also at 80 rows (lo pieces half of K at a time): GPU tests, same-box A/B at B=32 / B=16 against the build that fuses only <= 64 rows
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 5 1200 python -X faulthandler -m pytest tests/test_hip_round5.py tests/test_hip_round6.py tests/test_hip_parity.py -m gpu -q -x 2>&1 | tail -4
for b in 32 16; do for lib in new prev new prev; do
  if [ $lib = prev ]; then export A2P_NO_FUSED_FINAL=1; else unset A2P_NO_FUSED_FINAL; fi
  timeout -k 5 300 python bench.py --batch $b --no-cpu-baseline --no-parity --no-legs --steps 60 --warmup 8 > $O/r06_c71.json 2>/dev/null
  python - <<PY
import json
j=json.loads([l for l in open("$O/r06_c71.json") if l.startswith("{")][-1])
k=j["kernels"]; sub=k["_sub_classes"]
print("B=$b final_layer=$lib", j["value"], "steps/s", {a:v["avg_launch_us"] for a,v in sub.items()}, {x:(k[x]["avg_launch_us"],k[x]["launches_per_step"]) for x in ("gemm","attn_self","attn_cross")}, "decoder", j.get("decoder_mfma_frac"), "family", j["roofline"].get("chain_family"))
PY
done; done | sed 's/=new/=fused/; s/=prev/=launches/' | tee $O/r06_fused_final_80row_ab.txt
