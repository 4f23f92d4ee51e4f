#!/bin/bash
# round 6, call 34: attn3 rule extended to the body model's cross attention (head_dim 32, 2000 keys): body-model GPU tests, then the body leg alone, A/B by A2P_ATTN3=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests -m gpu -q -k "pose or body or attn3 or chain_vs or sampling_chain" 2>&1 | tail -4
for a in 1 0 1 0; do
  A2P_ATTN3=$a timeout -k 5 300 python bench.py --model pose --batch 16 --no-cpu-baseline --no-parity --no-legs --steps 100 --warmup 10 > $O/r06_c34_body_$a.json 2> $O/r06_c34_body_$a.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/r06_c34_body_$a.json") if l.startswith("{")][-1])
    k=j["kernels"]
    print("A2P_ATTN3=$a body B=16", j["value"], "steps/s", {a:k[a]["avg_launch_us"] for a in k if isinstance(k[a],dict) and "avg_launch_us" in k[a]})
except Exception as e:
    print("FAILED", e); print(open("$O/r06_c34_body_$a.err").read()[-800:])
PY
done | tee $O/r06_body_attn3_ab.txt
