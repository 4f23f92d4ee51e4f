#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
B="python bench.py --model pose --batch 16 --frames 600 --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-legs --no-kernel-timing"
for v in mt1 mt2 mt1b; do
  case $v in mt2) export A2P_GEMM_MT1=0;; *) export A2P_GEMM_MT1=1;; esac
  timeout -k 5 200 $B > gpurun_out/b19_body_$v.json 2> gpurun_out/b19_body_$v.err
  python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/b19_body_$v.json").read().strip().splitlines()[-1])
    print("$v", r["value"], r["ms_per_step"])
except Exception as e:
    print("$v failed", e); print(open("gpurun_out/b19_body_$v.err").read()[-1500:])
PY
done
unset A2P_GEMM_MT1
timeout -k 5 600 python -m pytest tests -m gpu -q -x -k "pose" > gpurun_out/b19_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/b19_tests.log
