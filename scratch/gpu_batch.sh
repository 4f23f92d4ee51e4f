#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
B="python bench.py --steps 100 --warmup 10 --repeats 2 --no-cpu-baseline --no-parity --no-legs"
run() {
  local tag=$1; shift
  timeout -k 5 120 env "$@" > gpurun_out/b25_$tag.json 2> gpurun_out/b25_$tag.err
  python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/b25_$tag.json").read().strip().splitlines()[-1])
    print("$tag", r["value"], r["ms_per_step"], {k: x["avg_launch_us"] for k, x in r.get("kernels", {}).items() if isinstance(x, dict)})
except Exception as e:
    print("$tag failed", e); print(open("gpurun_out/b25_$tag.err").read()[-600:])
PY
}
run w4 A2P_X=0 $B
run w8 A2P_ATTN_WAVES=8 $B
run w8occ A2P_ATTN_WAVES=9 $B
A2P_ATTN_WAVES=8 timeout -k 5 120 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "attention_kernel" 2>&1 | tail -2
