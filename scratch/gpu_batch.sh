#!/bin/bash
# The command list of the current gpurun call (one evolving script; git history keeps the earlier lists).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_round2.py tests/test_hip_parity.py -m gpu -x -q -k "T600 or golden or smoke or weight_updates" > gpurun_out/b6_tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/b6_tests.log
timeout 900 python bench.py --write-parity gpurun_out/b6_parity.json > gpurun_out/b6_bench.json 2> gpurun_out/b6_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/b6_bench.json"))
print("value", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["kernels"]["gemm"])
print("parity bar", d["parity"]["bar"])
for k, v in d["legs"].items():
    print(k, v["value"], v["roofline"]["kernel"], v["roofline"]["frac"], v["decoder_mfma_frac"], v["kernels"].get("gemm"))
PY
grep -h "T600" gpurun_out/parity_tests.json | head -0
python - <<'PY'
import json
d = json.load(open("gpurun_out/parity_tests.json"))
for k, v in d.items():
    if k.startswith("T600"): print(k, v)
PY
