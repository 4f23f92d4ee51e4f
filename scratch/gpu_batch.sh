#!/bin/bash
# round 4, call 2: generation-3 chain kernels after the PRE fix: bit-identity, phase shift / x prefetch sweep (isolated + in-step)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
T=c2
timeout -k 5 600 python -m pytest tests/test_hip_round3.py -m gpu -q -x -k "generation3" 2>&1 | tail -5 | tee gpurun_out/${T}_tests.log
for ph in 0 25 40; do for xpf in 0 1; do
  timeout -k 5 120 ./scratch/chain3_bench $ph $xpf 2>&1 | grep -A3 "M=38400" | grep -A2 "gen 3 rows 48 mode 2" | head -3
done; done | tee gpurun_out/${T}_chain3_sweep.txt
timeout -k 5 120 ./scratch/chain3_bench 30 1 > gpurun_out/${T}_chain3_bench.txt 2>&1
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
run b32_v3_p0 A2P_CHAIN_V=3 $B --batch 32
run b32_v3_p30x A2P_CHAIN_V=3 A2P_C3_PHASE=30 A2P_C3_XPF=1 $B --batch 32
run b32_v3_p0x A2P_CHAIN_V=3 A2P_C3_XPF=1 $B --batch 32
run b16_v1 A2P_X=0 $B --batch 16
run b16_v3_p30x A2P_CHAIN_V=3 A2P_C3_PHASE=30 A2P_C3_XPF=1 $B --batch 16
run body_v1 A2P_X=0 $B --model pose --batch 16
run body_v3_p15x A2P_CHAIN_V=3 A2P_C3_PHASE=15 A2P_C3_XPF=1 $B --model pose --batch 16
