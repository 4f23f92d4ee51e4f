#!/bin/bash
# round 4, call 14: price of the two round-4 additions to the attention tile body (same box, A2P_LIB_F16 A/B), alternating
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
T=c14
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
for i in 1 2; do
run b8_new$i A2P_X=0 $B --batch 8
run b8_r3attn$i A2P_LIB_F16=$PWD/scratch/ab/liba2p_attn_r3_f16.so $B --batch 8
done
run b32_new A2P_X=0 $B --batch 32
run b32_r3attn A2P_LIB_F16=$PWD/scratch/ab/liba2p_attn_r3_f16.so $B --batch 32
