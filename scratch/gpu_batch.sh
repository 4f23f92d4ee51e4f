#!/bin/bash
# round 6, call 58: attn3 rule refined for small batches (attn_kernel while its grid is one round): attention tests, then B = 1 / 2 / 4 / 8 lines, A/B by A2P_ATTN3=2 (attn3 wherever legal = the old rule at these sizes)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_hip_round6.py tests/test_hip_round4.py -m gpu -q -x 2>&1 | tail -3
for b in 1 2 4 8; do for a in 1 2; do
  A2P_ATTN3=$a timeout -k 5 300 python bench.py --batch $b --no-cpu-baseline --no-parity --no-legs --steps 60 --warmup 10 > $O/r06_c58.json 2>/dev/null
  python - <<PY
import json
j=json.loads([l for l in open("$O/r06_c58.json") if l.startswith("{")][-1])
k=j["kernels"]
print("B=$b A2P_ATTN3=$a", j["value"], "steps/s", {x:k[x]["avg_launch_us"] for x in k if isinstance(k[x],dict) and "avg_launch_us" in k[x]})
PY
done; done | tee $O/r06_attn3_rule_small_batch_ab.txt
