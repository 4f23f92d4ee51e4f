#!/bin/bash
# round 5, call 51: attn_kernel<half,64> with s_setprio(1) around its MFMA clusters (compile-time variant, A2P_ATTN_PRIO=0|1 forces) -- same-box A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 5 600 python -m pytest tests/test_hip_parity.py -m gpu -q -k "attention" > $O/r05_c51_tests.log 2>&1; tail -2 $O/r05_c51_tests.log
for b in 8 16; do for pr in 0 1 0 1; do
  A2P_ATTN_PRIO=$pr timeout -k 5 300 python bench.py --batch $b --no-cpu-baseline --no-parity --no-legs --steps 100 --warmup 10 > $O/r05_c51_b${b}_p$pr.json 2> $O/r05_c51_b${b}_p$pr.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/r05_c51_b${b}_p$pr.json") if l.startswith("{")][-1])
    k=j["kernels"]
    print("B=$b prio=$pr", j["value"], "attn self/cross us", k["attn_self"]["avg_launch_us"], k["attn_cross"]["avg_launch_us"], "chain", k["chain"]["avg_launch_us"])
except Exception as e:
    print("B=$b prio=$pr FAILED", e); print(open("$O/r05_c51_b${b}_p$pr.err").read()[-1200:])
PY
done; done 2>&1 | tee $O/r05_c51_ab.txt
