#!/bin/bash
# round 6, call 36: library with attn3_kernel v6: GPU tests that touch attention, then the headline / B=32 / body lines
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_hip_round6.py tests/test_hip_parity.py tests/test_hip_round2.py -m gpu -q -x 2>&1 | tail -3
for cfg in "--batch 8" "--batch 32 --steps 40" "--model pose --batch 16"; do
  timeout -k 5 400 python bench.py $cfg --no-cpu-baseline --no-parity --no-legs --warmup 10 > $O/r06_c36.json 2> $O/r06_c36.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/r06_c36.json") if l.startswith("{")][-1])
    k=j["kernels"]
    print("$cfg", j["value"], "steps/s", {a:k[a]["avg_launch_us"] for a in k if isinstance(k[a],dict) and "avg_launch_us" in k[a]}, "decoder", j.get("decoder_mfma_frac"))
except Exception as e:
    print("$cfg FAILED", e); print(open("$O/r06_c36.err").read()[-800:])
PY
done
