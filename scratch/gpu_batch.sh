#!/bin/bash
# round 4, call 1: vendor GEMM yardstick, generation-3 chain kernels (isolated bench, bit-identity tests, in-step A/B)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
T=c1
( timeout -k 5 200 python scratch/vendor_gemm.py > gpurun_out/${T}_vendor_gemm.txt 2> gpurun_out/${T}_vendor_gemm.err; echo "vendor rc=$?" )
( timeout -k 5 120 ./scratch/chain3_bench > gpurun_out/${T}_chain3_bench.txt 2>&1; echo "chain3_bench rc=$?"; tail -40 gpurun_out/${T}_chain3_bench.txt )
timeout -k 5 600 python -m pytest tests/test_hip_round3.py -m gpu -q -x -k "generation3" 2>&1 | tail -8 | tee gpurun_out/${T}_tests.log
B="python bench.py --steps 60 --warmup 10 --repeats 2 --no-cpu-baseline --no-parity --no-legs"
run() {
  local tag=$1; shift
  timeout -k 5 240 env "$@" > gpurun_out/${T}_$tag.json 2> gpurun_out/${T}_$tag.err
  python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/${T}_$tag.json").read().strip().splitlines()[-1])
    print("$tag", r["value"], r["ms_per_step"], {k: x["avg_launch_us"] for k, x in r.get("kernels", {}).items() if isinstance(x, dict)})
except Exception as e:
    print("$tag failed", e); print(open("gpurun_out/${T}_$tag.err").read()[-800:])
PY
}
run b32_v1 A2P_X=0 $B --batch 32
run b32_v3 A2P_CHAIN_V=3 $B --batch 32
run b8_v1 A2P_X=0 $B --batch 8
run b8_v3 A2P_CHAIN_V=3 $B --batch 8
run b16_v3 A2P_CHAIN_V=3 $B --batch 16
run b16_v1 A2P_X=0 $B --batch 16
run body_v1 A2P_X=0 $B --model pose --batch 16
run body_v3 A2P_CHAIN_V=3 $B --model pose --batch 16
