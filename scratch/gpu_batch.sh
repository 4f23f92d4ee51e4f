#!/bin/bash
# The command list of the current gpurun call (one evolving script; git history keeps the earlier lists).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_round3.py tests/test_hip_parity.py -m gpu -x -q -k "small_forward or ddim10 or bf16_mode_error or edge_shapes" > gpurun_out/b8_tests.log 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/b8_tests.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/parity_tests.json"))
for k, v in d.items():
    if k.startswith("small_vs"): print(k, v)
PY
for ns in 0 1; do
  if [ $ns = 1 ]; then export A2P_NO_SMALL=1; else unset A2P_NO_SMALL; fi
  timeout 600 python - <<'PY'
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda:0")
case = bench.Case("face", 1, 240, "fp16", dev, [0], respacing="ddim10", sampler="ddim")
rec = bench.leg_record(case, 50, 5, 3)
print("NO_SMALL" if os.environ.get("A2P_NO_SMALL") else "SMALL", rec["value"], rec["ms_per_step"], {k: (v["launches_per_step"], v["avg_launch_us"]) for k, v in rec["kernels"].items() if isinstance(v, dict)})
PY
done
