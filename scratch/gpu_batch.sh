#!/bin/bash
# round 4, call 6: which stage of the body path is not reproducible; why NaN persists after a poisoned forward; f1 fairseq blocks
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
T=c6
for i in 1 2 3; do timeout -k 5 200 python scratch/pipeline_determinism.py --overlap 0 2>&1 | grep "^run" ; done | tee gpurun_out/${T}_determinism.txt
timeout -k 5 200 python scratch/pipeline_determinism.py --overlap 1 2>&1 | grep "^run" | tee -a gpurun_out/${T}_determinism.txt
timeout -k 5 200 python scratch/nan_persist_probe.py fp16 2>&1 | grep -v amdgpu | tee gpurun_out/${T}_nan.txt
timeout -k 5 200 python scratch/nan_persist_probe.py fp32 2>&1 | grep -v amdgpu | tee -a gpurun_out/${T}_nan.txt
timeout -k 5 900 python -m pytest tests/test_frontend_hip.py -m gpu -q -x 2>&1 | tail -25 | tee gpurun_out/${T}_frontend.log
