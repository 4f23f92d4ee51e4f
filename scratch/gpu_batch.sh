#!/bin/bash
# round 6, call 62: MID kernels and the 48-row POST kernel store their finished rows in front of the LayerNorm too: bit identity, same-box A/B at B=8 / B=32 / B=16
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_hip_round5.py tests/test_hip_round6.py tests/test_hip_round2.py -m gpu -q -x 2>&1 | tail -3
for b in 8 32 16; do for lib in new prev new prev; do
  if [ $lib = prev ]; then export A2P_LIB_F16=$R/scratch/ab/liba2p_prev_f16.so; else unset A2P_LIB_F16; fi
  st=60; [ $b = 8 ] && st=100
  timeout -k 5 300 python bench.py --batch $b --no-cpu-baseline --no-parity --no-legs --steps $st --warmup 8 > $O/r06_c62.json 2>/dev/null
  python - <<PY
import json
j=json.loads([l for l in open("$O/r06_c62.json") if l.startswith("{")][-1])
k=j["kernels"]; sub=k["_sub_classes"]
print("B=$b lib=$lib", j["value"], "steps/s", {a:v["avg_launch_us"] for a,v in sub.items()}, {x:k[x]["avg_launch_us"] for x in ("attn_self","attn_cross")}, "decoder", j.get("decoder_mfma_frac"))
PY
done; done | tee $O/r06_store_early3_ab.txt
