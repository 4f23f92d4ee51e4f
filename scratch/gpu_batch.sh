#!/bin/bash
# round 5, call 14: tall chain v3 (256-column hidden chunks up to 64 rows, FFN fully unrolled, default everywhere): tests + bench A/B + stamps
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_hip_round5.py tests/test_hip_round2.py tests/test_hip_parity.py -m gpu -q -x > $O/r05_c14_tests.log 2>&1; tail -3 $O/r05_c14_tests.log
for b in 8 16 32; do for v in 1 0 1 0; do
  A2P_CHAIN_V=$v timeout -k 5 300 python bench.py --batch $b --no-cpu-baseline --no-parity --no-legs --steps 60 --warmup 10 > $O/r05_c14_b${b}_v$v.json 2> $O/r05_c14_b${b}_v$v.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/r05_c14_b${b}_v$v.json") if l.startswith("{")][-1])
    k=j["kernels"]; sub=k.get("_sub_classes",{})
    print("B=$b V=$v", j["value"], "chain", k["chain"]["ms_per_step"], {n:(v["avg_launch_us"], v.get("mfma_frac")) for n,v in sub.items()})
except Exception as e:
    print("B=$b V=$v FAILED", e); print(open("$O/r05_c14_b${b}_v$v.err").read()[-1500:])
PY
done; done
export A2P_LIB_F16=$R/scratch/ab/liba2p_stamps_f16.so
for b in 8 32; do PP_BATCH=$b A2P_STAMP_LAUNCH=4 timeout -k 5 300 python scratch/phase_probe4.py 2>&1 | grep "gen 4"; done > $O/r05_c14_phase_probe4.txt
cat $O/r05_c14_phase_probe4.txt
