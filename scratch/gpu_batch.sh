#!/bin/bash
# round 6, call 55: final_layer's split operand rows stored by the last tall POST kernel (no split3_kernel launch): bit identity, chain tests, A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_hip_round6.py tests/test_hip_round5.py -m gpu -q -x -k "split_rows or full_sampling or benchmarked_batch or tall or bit" 2>&1 | tail -4
for f in 0 1 0 1; do
  A2P_NO_T3_FUSED=$f timeout -k 5 300 python bench.py --no-cpu-baseline --no-parity --no-legs --steps 100 --warmup 10 > $O/r06_c55.json 2> $O/r06_c55.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/r06_c55.json") if l.startswith("{")][-1])
    k=j["kernels"]; sub=k["_sub_classes"]
    print("A2P_NO_T3_FUSED=$f B=8", j["value"], "steps/s", j["ms_per_step"], "ms", {a:k[a]["avg_launch_us"] for a in ("chain","gemm","attn_self","attn_cross")}, {a:v["avg_launch_us"] for a,v in sub.items()})
except Exception as e:
    print("FAILED", e); print(open("$O/r06_c55.err").read()[-800:])
PY
done | tee $O/r06_t3_fused_ab.txt
