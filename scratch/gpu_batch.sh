#!/bin/bash
# The command list of the current gpurun call (one evolving script; git history keeps the earlier lists).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_round3.py -m gpu -x -q -k "generation2" > gpurun_out/b5_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/b5_tests.log
timeout 300 ./scratch/chain2_bench > gpurun_out/chain2_bench.txt 2>&1
grep -v "block " gpurun_out/chain2_bench.txt | grep -v "x 1 waves\|x 8 waves" | head -40
grep -A2 "leaders 8 x 2 waves\]  gen 2 rows 48 mode 2" gpurun_out/chain2_bench.txt | head -8
run() {  # name, env..., -- bench args
  name=$1; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-parity --no-legs $BARGS > gpurun_out/b5_$name.json 2> gpurun_out/b5_$name.err
  python - "$name" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/b5_{n}.json"))
    k = d["kernels"]
    print(f"{n:14s} {d['value']:8.2f} steps/s  {d['ms_per_step']:.4f} ms  chain {k['chain']['avg_launch_us']:7.2f} us frac {d['roofline']['frac']:.4f}  cross {k['attn_cross']['avg_launch_us']:.1f} self {k['attn_self']['avg_launch_us']:.1f}")
except Exception as e:
    print(n, "failed", e)
PY
}
BARGS="--batch 8"
run b8_v1 A2P_CHAIN_V=1
run b8_v2 A2P_CHAIN_V=2
run b8_v2_nolead A2P_CHAIN_V=2 A2P_CHAIN_LEADERS=0
run b8_v2_pfw4 A2P_CHAIN_V=2 A2P_CHAIN_PFW=4
BARGS="--batch 32 --steps 8 --warmup 2"
run b32_v1 A2P_CHAIN_V=1
run b32_v2 A2P_CHAIN_V=2
run b32_v2_mt4 A2P_CHAIN_V=2 A2P_CHAIN_MT=4
run b32_v2_mt3 A2P_CHAIN_V=2 A2P_CHAIN_MT=3
BARGS="--model pose --batch 16"
run pose_v1 A2P_CHAIN_V=1
run pose_v2 A2P_CHAIN_V=2
run pose_v2_nolead A2P_CHAIN_V=2 A2P_CHAIN_LEADERS=0
