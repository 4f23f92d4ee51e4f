#!/bin/bash
# round 4, call 13: attention without the s_setprio branches, placements test, full GPU suite
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
T=c13
B="python bench.py --steps 60 --warmup 10 --repeats 2 --no-cpu-baseline --no-parity --no-legs"
run() {
  local tag=$1; shift
  timeout -k 5 240 env "$@" > gpurun_out/${T}_$tag.json 2> gpurun_out/${T}_$tag.err
  python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/${T}_$tag.json").read().strip().splitlines()[-1])
    print("$tag", r["value"], r["ms_per_step"], {k: (x["avg_launch_us"], x["launches_per_step"]) for k, x in r.get("kernels", {}).items() if isinstance(x, dict)})
except Exception as e:
    print("$tag failed", e); print(open("gpurun_out/${T}_$tag.err").read()[-800:])
PY
}
run b8 A2P_X=0 $B --batch 8
run b32 A2P_X=0 $B --batch 32
timeout -k 5 900 python -m pytest tests/test_hip_round4.py -m gpu -q -x -k "placements" 2>&1 | grep -E "Differing|^E  |passed|failed" | head -20 | cut -c1-600 | tee gpurun_out/${T}_placements.log
timeout -k 5 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/${T}_tests_all.log
cp gpurun_out/parity_tests.json gpurun_out/${T}_parity_tests.json 2>/dev/null
