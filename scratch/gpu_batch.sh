#!/bin/bash
# round 5, call 34: final HEAD -- full GPU suite, smoke(), default bench line (-> profiles/r05_bench_default.json)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
( time timeout -k 5 1200 python -m pytest tests -m gpu -q -x ) > $O/r05_c34_tests.log 2>&1; tail -5 $O/r05_c34_tests.log
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 5 900 python bench.py > $O/r05_c34_bench_default.json 2> $O/r05_c34_bench_default.err; python - <<PY
import json
j=json.loads([l for l in open("$O/r05_c34_bench_default.json") if l.startswith("{")][-1])
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["chain_family"], j["decoder_mfma_frac"], {n:(l.get("value"), l.get("decoder_mfma_frac")) for n,l in j["legs"].items()})
print("under_load", j.get("under_load")); print("b32 under_load", j["legs"]["b32"].get("under_load"))
print({n:(v["avg_launch_us"], v.get("mfma_frac")) for n,v in j["legs"]["b32"]["kernels"]["_sub_classes"].items()})
print(json.dumps(j["box"])[:1800])
PY
