#!/bin/bash
# The command list of the current gpurun call (one evolving script; git history keeps the earlier lists).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 5 600 python bench.py --write-parity gpurun_out/f2_parity.json > gpurun_out/f2_bench.json 2> gpurun_out/f2_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    r = json.loads(open("gpurun_out/f2_bench.json").read().strip().splitlines()[-1])
    print("headline", r["value"], r["ms_per_step"], "roofline", r["roofline"]["frac"], "cpu", r["cpu_baseline"]["value"])
    for k, v in r.get("legs", {}).items(): print("  leg", k, v.get("value"), v.get("ms_per_step"), (v.get("roofline") or {}).get("kernel"), (v.get("roofline") or {}).get("frac"))
except Exception as e:
    print("bench parse failed", e)
PY
cd /tmp
prof() {  # tag, bench args
  local tag=$1; shift
  timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-kernel-timing --no-legs --no-parity --repeats 1 > $R/gpurun_out/prof_$tag.log 2>&1
  echo "prof $tag rc=$?"
  cp $R/gpurun_out/prof_$tag/p_kernel_stats.csv $R/gpurun_out/kernel_stats_$tag.csv 2>/dev/null
  rm -f $R/gpurun_out/prof_$tag/*kernel_trace.csv
  head -5 $R/gpurun_out/kernel_stats_$tag.csv | cut -c1-130
}
prof f2_b8 --steps 5 --warmup 2
prof f2_cfg0 --batch 1 --frames 240 --steps 20 --warmup 3
prof f2_body --model pose --batch 16 --steps 5 --warmup 2
prof f2_b32 --batch 32 --steps 3 --warmup 1
