#!/bin/bash
# round 6, call 79: 48-row MID kernel with its residual rows requested in the prologue burst (CHAIN4_MID_PREX=1, ring 8 deep) against the shipped one (ring 16 deep): bit identity, same-box A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
A2P_LIB_F16=$R/scratch/ab/liba2p_prex_f16.so timeout -k 5 900 python -m pytest tests/test_hip_round5.py -m gpu -q -x -k "fp16" 2>&1 | tail -2
for b in 8 4; do for lib in prex base prex base; do
  if [ $lib = prex ]; then export A2P_LIB_F16=$R/scratch/ab/liba2p_prex_f16.so; else unset A2P_LIB_F16; fi
  timeout -k 5 300 python bench.py --batch $b --no-cpu-baseline --no-parity --no-legs --steps 100 --warmup 8 > $O/r06_c79.json 2>/dev/null
  python - <<PY
import json
j=json.loads([l for l in open("$O/r06_c79.json") if l.startswith("{")][-1])
k=j["kernels"]; sub=k["_sub_classes"]
print("B=$b lib=$lib", j["value"], "steps/s", {a:v["avg_launch_us"] for a,v in sub.items()}, "family", j["roofline"].get("chain_family"))
PY
done; done | tee $O/r06_mid_prex_ab.txt
