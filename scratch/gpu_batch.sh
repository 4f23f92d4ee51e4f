#!/bin/bash
# round 5, calls 35+: one short B=8 / B=16 / B=32 line per box with the box and under-load records (profiles/r05_boxes.md)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
TAG=${1:-c35}
cd $R
for b in ${BATCHES:-8 16 32}; do
  timeout -k 5 300 python bench.py --batch $b --no-cpu-baseline --no-parity --no-legs --steps 100 --warmup 10 > $O/r05_${TAG}_b$b.json 2> $O/r05_${TAG}_b$b.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/r05_${TAG}_b$b.json") if l.startswith("{")][-1])
    sub=j["kernels"].get("_sub_classes",{}); ul=j.get("under_load") or {}; bx=j["box"]
    uid=[r for r in (bx.get("showhw") or []) if "Unique ID" in r]
    print("B=$b", j["value"], j["roofline"]["chain_family"], {n:v["avg_launch_us"] for n,v in sub.items()}, ul.get("power_w"), ul.get("sclk_mhz"), ul.get("junction_c"), uid)
except Exception as e:
    print("B=$b FAILED", e); print(open("$O/r05_${TAG}_b$b.err").read()[-800:])
PY
done
