#!/bin/bash
# round 6, call 53: attn3 v6 edge cases in the GPU test (reference move on the last tile, spike on the last tile's key 0)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 5 600 python -m pytest tests/test_hip_round6.py -m gpu -q -k "attn3_kernel_vs_fp64" 2>&1 | tail -15
python - <<'PY'
import json
d=json.load(open("gpurun_out/parity_tests.json")) if __import__("os").path.exists("gpurun_out/parity_tests.json") else {}
for k,v in d.items():
    if "spike_last" in k: print(k, v)
PY
