#!/bin/bash
# round 5, call 31: fp32 parity mode, attention at 2 waves per SIMD (no spills) vs 3 (42 spilled registers); under-load probe check
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
for lib in w3 w2 w3 w2; do
  if [ $lib = w2 ]; then export A2P_LIB=$R/scratch/ab/liba2p_f32w2.so; else unset A2P_LIB; fi
  timeout -k 5 300 python bench.py --precision fp32 --no-cpu-baseline --no-parity --no-legs --steps 40 --warmup 5 > $O/r05_c31_fp32_$lib.json 2> $O/r05_c31_fp32_$lib.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/r05_c31_fp32_$lib.json") if l.startswith("{")][-1])
    k=j["kernels"]
    print("fp32 lib=$lib", j["value"], {n:(v.get("ms_per_step"), v.get("avg_launch_us")) for n,v in k.items() if isinstance(v,dict) and "ms_per_step" in v}, "load", j.get("under_load"))
except Exception as e:
    print("fp32 lib=$lib FAILED", e); print(open("$O/r05_c31_fp32_$lib.err").read()[-1500:])
PY
done 2>&1 | tee $O/r05_c31_ab.txt
unset A2P_LIB
timeout -k 5 300 python bench.py --no-cpu-baseline --no-parity --no-legs --steps 100 --warmup 10 > $O/r05_c31_b8.json 2> $O/r05_c31_b8.err
python - <<PY
import json
j=json.loads([l for l in open("$O/r05_c31_b8.json") if l.startswith("{")][-1])
print("B=8", j["value"], "load", j.get("under_load"))
PY
