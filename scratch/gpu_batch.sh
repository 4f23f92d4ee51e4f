#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
B="python bench.py --batch 1 --frames 240 --steps 50 --warmup 5 --no-cpu-baseline --no-parity --no-legs"
for v in default; do
  timeout -k 5 200 $B > gpurun_out/b17_cfg0_$v.json 2> gpurun_out/b17_cfg0_$v.err
  python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/b17_cfg0_$v.json").read().strip().splitlines()[-1])
    print("$v", r["value"], r["ms_per_step"])
    for k, x in r.get("kernels", {}).items():
        if isinstance(x, dict) and "avg_launch_us" in x: print("   ", k, x.get("launches_per_step"), x["avg_launch_us"])
except Exception as e:
    print("$v failed", e)
PY
done
timeout -k 5 1500 python -m pytest tests -m gpu -q -x > gpurun_out/b17_tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/b17_tests.log
cp gpurun_out/parity_tests.json gpurun_out/b17_parity_tests.json 2>/dev/null
