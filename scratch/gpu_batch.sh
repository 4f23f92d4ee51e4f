#!/bin/bash
# round 6, call 54: the full chains gated against the REFERENCE's own chain states (tests/golden/golden_chain_*_ref_v1.npz)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 5 600 python -m pytest tests/test_hip_round6.py -m gpu -q -k "full_sampling_chain" 2>&1 | tail -3
python - <<'PY'
import json
d=json.load(open("gpurun_out/parity_tests.json"))
for k,v in d.items():
    if k.startswith("chain_vs_reference"): print(k, {a:b for a,b in v.items() if a.endswith(("step1000","step100"))})
PY
