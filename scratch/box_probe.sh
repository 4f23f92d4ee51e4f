#!/bin/bash
# One line per gpurun box: headline bench (short) + what rocm-smi says about the box (VERDICT weak 8: box-to-box spread)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
TAG=$1
python bench.py --steps 200 --warmup 3 --repeats 2 --no-cpu-baseline --no-parity --no-legs > gpurun_out/box_$TAG.json 2> gpurun_out/box_$TAG.err &
BP=$!
: > gpurun_out/box_$TAG.smi_during.txt
while kill -0 $BP 2>/dev/null; do   # one sample per ~0.5 s while the bench runs
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Package Power|sclk" | tr -s '\t ' ' ' | tr '\n' ' ' >> gpurun_out/box_$TAG.smi_during.txt
  echo >> gpurun_out/box_$TAG.smi_during.txt
  sleep 0.3
done
wait $BP
{
  python - <<PY
import json
r = json.loads(open("gpurun_out/box_$TAG.json").read().strip().splitlines()[-1])
k = r["kernels"]
print("box $TAG: %.1f steps/s  chain %.1f us  attn_cross %.1f us  attn_self %.1f us  (chain shape: %s)" % (r["value"], k["chain"]["avg_launch_us"], k["attn_cross"]["avg_launch_us"], k["attn_self"]["avg_launch_us"], r.get("config", {}).get("chain_shape", "?")))
PY
  echo "  during the run (samples with the highest sclk):"; awk '{m=$0; sub(/.*\(/,"",m) } { if (match($0, /\(([0-9]+)Mhz\)/)) { v=substr($0, RSTART+1, RLENGTH-5); print v, $0 } }' gpurun_out/box_$TAG.smi_during.txt | sort -n -r | head -3 | cut -d' ' -f2- | sed 's/^/    /'
  echo "    samples: $(grep -c sclk gpurun_out/box_$TAG.smi_during.txt)"
  echo "  idle:"; rocm-smi --showmaxpower --showuniqueid --showvbios --showperflevel 2>/dev/null | grep -E "Max Graphics|Unique ID|VBIOS|Performance Level" | sed 's/^/    /'
  uname -r | sed 's/^/    kernel /'
} | tee gpurun_out/box_$TAG.txt
