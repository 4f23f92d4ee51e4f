#!/bin/bash
# round 5, call 25: asm fragment reads only in the 80-row POST kernel: bit-identity tests, then same-box A/B vs the f90d4d4 lib
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 5 600 python -m pytest tests/test_hip_round5.py -m gpu -q > $O/r05_c25_tests.log 2>&1; grep -E "passed|failed|FAILED|AssertionError: " $O/r05_c25_tests.log | head -20
for b in 8 16 32; do for lib in head new head new; do
  if [ $lib = head ]; then export A2P_LIB_F16=$R/scratch/ab/liba2p_head_f16.so; else unset A2P_LIB_F16; fi
  A2P_CHAIN_V=4 timeout -k 5 300 python bench.py --batch $b --no-cpu-baseline --no-parity --no-legs --steps 60 --warmup 10 > $O/r05_c25_b${b}_$lib.json 2> $O/r05_c25_b${b}_$lib.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/r05_c25_b${b}_$lib.json") if l.startswith("{")][-1])
    k=j["kernels"]; sub=k.get("_sub_classes",{})
    print("B=$b lib=$lib", j["value"], "chain", k["chain"]["ms_per_step"], {n:(v["avg_launch_us"], v.get("mfma_frac")) for n,v in sub.items()})
except Exception as e:
    print("B=$b lib=$lib FAILED", e); print(open("$O/r05_c25_b${b}_$lib.err").read()[-1500:])
PY
done; done 2>&1 | tee $O/r05_c25_ab.txt
