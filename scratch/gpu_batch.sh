#!/bin/bash
# round 6, call 77: rotary entries of all four tiles requested up front in the 48-row kernels (CHAIN4_CS_ALL): bit identity, same-box A/B at B=8 / B=4
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_hip_round5.py tests/test_hip_round6.py -m gpu -q -x -k "fp16 and (tall or inside)" 2>&1 | tail -2
for b in 8 4; do for lib in new prev new prev; do
  if [ $lib = prev ]; then export A2P_LIB_F16=$R/scratch/ab/liba2p_s4_f16.so; else unset A2P_LIB_F16; fi
  timeout -k 5 300 python bench.py --batch $b --no-cpu-baseline --no-parity --no-legs --steps 100 --warmup 8 > $O/r06_c77.json 2>/dev/null
  python - <<PY
import json
j=json.loads([l for l in open("$O/r06_c77.json") if l.startswith("{")][-1])
k=j["kernels"]; sub=k["_sub_classes"]
print("B=$b lib=$lib", j["value"], "steps/s", {a:v["avg_launch_us"] for a,v in sub.items()}, "family", j["roofline"].get("chain_family"))
PY
done; done | tee $O/r06_cs_all_ab.txt
