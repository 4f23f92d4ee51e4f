#!/bin/bash
# The command list of the current gpurun call (one evolving script; git history keeps the earlier lists).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
python __graft_entry__.py --smoke > gpurun_out/b1_smoke.log 2>&1; echo "smoke rc=$?" 
timeout 900 python -m pytest tests/test_hip_round2.py -m gpu -x -q -k "T600 or workgroup_shapes or mixed_panel" > gpurun_out/b1_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/b1_tests.log
timeout 900 python bench.py --write-parity gpurun_out/b1_parity.json > gpurun_out/b1_bench.json 2> gpurun_out/b1_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/b1_bench.json"))
print("value", d["value"], d["ms_per_step"], d["roofline"]["frac"])
print("parity bar", d["parity"]["bar"])
for k, v in d["legs"].items():
    print(k, v["value"], v["roofline"]["kernel"], v["roofline"]["frac"], v["decoder_mfma_frac"])
PY
