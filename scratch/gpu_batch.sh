#!/bin/bash
# round 5, call 9: tall chain kernels (kernels_chain4.h): bit-identity tests + bench A/B at B = 8 / 16 / 32
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_hip_round5.py -m gpu -q -x > $O/r05_c9_tests.log 2>&1; tail -5 $O/r05_c9_tests.log
for b in 32 16 8; do for v in 1 4 1 4; do
  A2P_CHAIN_V=$v timeout -k 5 300 python bench.py --batch $b --no-cpu-baseline --no-parity --no-legs --steps 40 --warmup 5 > $O/r05_c9_b${b}_v$v.json 2> $O/r05_c9_b${b}_v$v.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/r05_c9_b${b}_v$v.json") if l.startswith("{")][-1])
    k=j["kernels"]; sub=k.get("_sub_classes",{})
    print("B=$b V=$v", j["value"], "chain", k["chain"]["ms_per_step"], {n:(v["avg_launch_us"], v.get("mfma_frac")) for n,v in sub.items()})
except Exception as e:
    print("B=$b V=$v FAILED", e); print(open("$O/r05_c9_b${b}_v$v.err").read()[-1500:])
PY
done; done
