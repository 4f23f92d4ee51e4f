#!/bin/bash
# The command list of the current gpurun call (one evolving script; git history keeps the earlier lists).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_round3.py tests/test_hip_parity.py -m gpu -x -q -k "small_forward or bf16_mode_error" > gpurun_out/b9_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/b9_tests.log
for ns in 0 1; do
  if [ $ns = 1 ]; then export A2P_NO_SMALL=1; else unset A2P_NO_SMALL; fi
  timeout 600 python - <<'PY'
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda:0")
case = bench.Case("face", 1, 240, "fp16", dev, [0], respacing="ddim10", sampler="ddim")
rec = bench.leg_record(case, 50, 5, 3)
print("NO_SMALL" if os.environ.get("A2P_NO_SMALL") else "SMALL", rec["value"], rec["ms_per_step"], {k: (v["launches_per_step"], v["avg_launch_us"]) for k, v in rec["kernels"].items() if isinstance(v, dict)})
PY
done
unset A2P_NO_SMALL
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg0_r03 -o p -- python $R/bench.py --batch 1 --frames 240 --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-legs --repeats 1 > $R/gpurun_out/prof_cfg0_r03.log 2>&1
cp $R/gpurun_out/prof_cfg0_r03/p_kernel_stats.csv $R/gpurun_out/kernel_stats_cfg0_r03.csv; rm -f $R/gpurun_out/prof_cfg0_r03/*kernel_trace.csv
head -8 $R/gpurun_out/kernel_stats_cfg0_r03.csv | cut -c1-140
