#!/bin/bash
# round 4, call 23: 128-row blocks of the fused body tail: parity against the nine GEMM launches, then the body bench leg
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout -k 5 600 python -m pytest tests/test_hip_round4.py -x -q -m gpu -k "fused_pose_tail or island_ab" 2>&1 | tail -6
timeout -k 5 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "pose" 2>&1 | tail -3
for i in 1 2; do
timeout -k 5 300 python bench.py --model pose --batch 16 --steps 100 --warmup 10 --repeats 2 --no-cpu-baseline --no-parity --no-legs > gpurun_out/c23_body_$i.json 2> gpurun_out/c23_body_$i.err
python - <<PY
import json
r = json.loads(open("gpurun_out/c23_body_$i.json").read().strip().splitlines()[-1])
print("body", r["value"], r["ms_per_step"], {k: (x["avg_launch_us"], x["launches_per_step"]) for k, x in r.get("kernels", {}).items() if isinstance(x, dict)})
PY
done
