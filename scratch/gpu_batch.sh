#!/bin/bash
# round 5, call 29: same A/B as call 28 at B=8 (hoping for a fast-type box) + the under-load power / clock record of bench.py
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
for b in 8; do for lib in head new asmall head new asmall; do
  if [ $lib = head ]; then export A2P_LIB_F16=$R/scratch/ab/liba2p_head_f16.so; elif [ $lib = asmall ]; then export A2P_LIB_F16=$R/scratch/ab/liba2p_asmall_f16.so; else unset A2P_LIB_F16; fi
  A2P_CHAIN_V=4 timeout -k 5 300 python bench.py --batch $b --no-cpu-baseline --no-parity --no-legs --steps 100 --warmup 10 > $O/r05_c29_b${b}_$lib.json 2> $O/r05_c29_b${b}_$lib.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/r05_c29_b${b}_$lib.json") if l.startswith("{")][-1])
    k=j["kernels"]; sub=k.get("_sub_classes",{})
    print("B=$b lib=$lib", j["value"], "chain", k["chain"]["ms_per_step"], {n:(v["avg_launch_us"], v.get("mfma_frac")) for n,v in sub.items()}, "attn", k["attn_self"]["avg_launch_us"], k["attn_cross"]["avg_launch_us"], "load", j.get("under_load"))
except Exception as e:
    print("B=$b lib=$lib FAILED", e); print(open("$O/r05_c29_b${b}_$lib.err").read()[-1500:])
PY
done; done 2>&1 | tee $O/r05_c29_ab.txt
unset A2P_LIB_F16
timeout -k 5 300 python bench.py --batch 32 --no-cpu-baseline --no-parity --no-legs --steps 60 --warmup 10 > $O/r05_c29_b32_auto.json 2> $O/r05_c29_b32_auto.err
python - <<PY
import json
j=json.loads([l for l in open("$O/r05_c29_b32_auto.json") if l.startswith("{")][-1])
print("B=32 auto", j["value"], j["roofline"]["chain_family"], {n:v["avg_launch_us"] for n,v in j["kernels"]["_sub_classes"].items()}, "load", j.get("under_load"))
print(json.dumps(j["box"])[:400])
PY
