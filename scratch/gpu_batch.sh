#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 5 400 python -m pytest tests/test_hip_round3.py -m gpu -q -x -k "small_forward or captured" > gpurun_out/b20_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/b20_tests.log
B="python bench.py --batch 1 --frames 240 --steps 50 --warmup 5 --no-cpu-baseline --no-parity --no-legs"
for v in nw8 nw4 nw8b; do
  unset A2P_SMALL_NW
  case $v in nw4) export A2P_SMALL_NW=4;; esac
  timeout -k 5 200 $B > gpurun_out/b20_cfg0_$v.json 2> gpurun_out/b20_cfg0_$v.err
  python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/b20_cfg0_$v.json").read().strip().splitlines()[-1])
    print("$v", r["value"], r["ms_per_step"], {k: x["avg_launch_us"] for k, x in r.get("kernels", {}).items() if isinstance(x, dict)})
except Exception as e:
    print("$v failed", e); print(open("gpurun_out/b20_cfg0_$v.err").read()[-1500:])
PY
done
