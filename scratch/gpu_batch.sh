#!/bin/bash
# The command list of the current gpurun call (one evolving script; git history keeps the earlier lists).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-f7}
t0=$(date +%s)
timeout -k 5 900 python bench.py --write-parity gpurun_out/${TAG}_parity.json > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"
python - <<PY
import json
try:
    r = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
    print("headline", r["value"], r["ms_per_step"], r["steps"], r["warmup"], "roofline", r["roofline"]["frac"], "cpu", r["cpu_baseline"]["value"], r["cpu_baseline"].get("reference_as_is", {}))
    for k, v in r.get("legs", {}).items(): print("  leg", k, v.get("value"), v.get("ms_per_step"), (v.get("roofline") or {}).get("kernel"), (v.get("roofline") or {}).get("frac"))
    print("parity bar", r["parity"]["bar"])
except Exception as e:
    print("bench parse failed", e)
PY
timeout -k 5 800 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${TAG}_tests.log | cut -c1-160
cp gpurun_out/parity_tests.json gpurun_out/${TAG}_parity_tests.json
