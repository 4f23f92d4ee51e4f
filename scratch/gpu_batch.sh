#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity --no-legs --no-kernel-timing"
run() {
  local tag=$1; shift
  timeout -k 5 200 env "$@" > gpurun_out/b24_$tag.json 2> gpurun_out/b24_$tag.err
  python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/b24_$tag.json").read().strip().splitlines()[-1])
    print("$tag", r["value"], r["ms_per_step"])
except Exception as e:
    print("$tag failed", e); print(open("gpurun_out/b24_$tag.err").read()[-600:])
PY
}
run r1200_small A2P_X=0 $B --batch 1 --frames 600
run r1200_chain A2P_CHAIN_ROWS=1 $B --batch 1 --frames 600
run r1440_small A2P_CHAIN_ROWS=99999 $B --batch 3 --frames 240
run r1440_chain A2P_X=0 $B --batch 3 --frames 240
run r1920_small A2P_CHAIN_ROWS=99999 $B --batch 4 --frames 240
run r1920_chain A2P_X=0 $B --batch 4 --frames 240
run r2400_small A2P_CHAIN_ROWS=99999 $B --batch 2 --frames 600
run r2400_chain A2P_X=0 $B --batch 2 --frames 600
