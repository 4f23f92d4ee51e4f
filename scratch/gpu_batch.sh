#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-legs"
run() {  # tag, env, args
  local tag=$1; shift
  timeout -k 5 200 env "$@" > gpurun_out/b23_$tag.json 2> gpurun_out/b23_$tag.err
  python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/b23_$tag.json").read().strip().splitlines()[-1])
    print("$tag", r["value"], r["ms_per_step"], {k: x["avg_launch_us"] for k, x in r.get("kernels", {}).items() if isinstance(x, dict)})
except Exception as e:
    print("$tag failed", e); print(open("gpurun_out/b23_$tag.err").read()[-800:])
PY
}
run b8_qt2 A2P_X=0 $B
run b8_qt1_w4 A2P_ATTN_QT=1 $B
run b8_qt1_w8 A2P_ATTN_QT=1 A2P_ATTN_WAVES=8 $B
run b32_qt2 A2P_X=0 $B --batch 32 --steps 8
run b32_qt1_w8 A2P_ATTN_QT=1 A2P_ATTN_WAVES=8 $B --batch 32 --steps 8
A2P_ATTN_QT=1 A2P_ATTN_WAVES=8 timeout -k 5 300 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "attention_kernel or denoiser or forward" > gpurun_out/b23_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/b23_tests.log
