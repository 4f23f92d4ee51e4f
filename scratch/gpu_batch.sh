#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
A2P_LIB_F16=scratch/ab/liba2p_stamps_f16.so timeout 600 python scratch/phase_probe.py > gpurun_out/phase_probe_b8.txt 2>&1; echo "rc=$?"
cat gpurun_out/phase_probe_b8.txt | grep -v amdgpu.ids
PP_BATCH=32 A2P_LIB_F16=scratch/ab/liba2p_stamps_f16.so timeout 600 python scratch/phase_probe.py > gpurun_out/phase_probe_b32.txt 2>&1; echo "rc=$?"
cat gpurun_out/phase_probe_b32.txt | grep -v amdgpu.ids
