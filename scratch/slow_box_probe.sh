#!/bin/bash
# find a slow-type GPU (PRE > 40 us at B=8) and time the latency-tolerant candidates there.  On a fast-type box the script stops after one short bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
line() {  # $1 = tag, rest = env + bench args
  local tag=$1; shift
  env "$@" timeout -k 5 300 python bench.py --no-cpu-baseline --no-parity --no-legs --warmup 10 $BARGS > $O/slow_probe.json 2> $O/slow_probe.err
  python - "$tag" <<PY
import json, sys
try:
    j=json.loads([l for l in open("$O/slow_probe.json") if l.startswith("{")][-1])
    sub=j["kernels"]["_sub_classes"]; k=j["kernels"]
    uid=[r for r in j["box"]["showhw"] if "Unique" in r][-1:]
    print(sys.argv[1], j["value"], "steps/s", {n:v["avg_launch_us"] for n,v in sub.items()}, {a:k[a]["avg_launch_us"] for a in ("attn_self","attn_cross")}, "family", j["roofline"].get("chain_family"), uid)
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
BARGS="--batch 8 --steps 60"
line "probe B=8 default" A2P_X=0 | tee $O/slow_probe_first.txt
PRE=$(python - <<PY
import re
s=open("$O/slow_probe_first.txt").read()
m=re.search(r"'chain_pre': ([0-9.]+)", s); print(m.group(1) if m else 0)
PY
)
echo "chain_pre = $PRE us"
if python -c "import sys; sys.exit(0 if float('$PRE') > 40.0 else 1)"; then
  echo "SLOW-TYPE BOX: running the A/B" | tee $O/r06_slow_box_ab.txt
  for rep in 1 2; do
    BARGS="--batch 8 --steps 100"
    line "B=8 gen1 (V=1)" A2P_CHAIN_V=1 | tee -a $O/r06_slow_box_ab.txt
    line "B=8 tall (V=4)" A2P_CHAIN_V=4 | tee -a $O/r06_slow_box_ab.txt
    line "B=8 tall POST parked + 16-deep ring (V=4)" A2P_CHAIN_V=4 A2P_LIB_F16=$R/scratch/ab/liba2p_park3_f16.so | tee -a $O/r06_slow_box_ab.txt
    line "B=8 default (calibrated)" A2P_TUNE_VERBOSE=0 | tee -a $O/r06_slow_box_ab.txt
    BARGS="--batch 32 --steps 40"
    line "B=32 gen1 (V=1)" A2P_CHAIN_V=1 | tee -a $O/r06_slow_box_ab.txt
    line "B=32 tall (V=4)" A2P_CHAIN_V=4 | tee -a $O/r06_slow_box_ab.txt
    line "B=32 default (calibrated)" A2P_TUNE_VERBOSE=0 | tee -a $O/r06_slow_box_ab.txt
  done
else
  echo "fast-type box: nothing to do"
fi
