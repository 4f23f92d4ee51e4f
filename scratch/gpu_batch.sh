#!/bin/bash
# round 5, call 33: 48-row MID kernel with the 16-deep ring in the library: bit-identity tests, smoke(), same-box B=8 A/B vs the previous build
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 5 600 python -m pytest tests/test_hip_round5.py -m gpu -q > $O/r05_c33_tests.log 2>&1; grep -E "passed|failed|FAILED|AssertionError: " $O/r05_c33_tests.log | head
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for lib in prev new prev new; do
  if [ $lib = prev ]; then export A2P_LIB_F16=$R/scratch/ab/liba2p_prev_f16.so; else unset A2P_LIB_F16; fi
  timeout -k 5 300 python bench.py --batch 8 --no-cpu-baseline --no-parity --no-legs --steps 100 --warmup 10 > $O/r05_c33_b8_$lib.json 2> $O/r05_c33_b8_$lib.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/r05_c33_b8_$lib.json") if l.startswith("{")][-1])
    k=j["kernels"]; sub=k.get("_sub_classes",{})
    print("B=8 lib=$lib", j["value"], j["roofline"]["chain_family"], {n:v["avg_launch_us"] for n,v in sub.items()}, (j.get("under_load") or {}).get("power_w"), (j.get("under_load") or {}).get("sclk_mhz"))
except Exception as e:
    print("B=8 lib=$lib FAILED", e); print(open("$O/r05_c33_b8_$lib.err").read()[-1500:])
PY
done 2>&1 | tee $O/r05_c33_ab.txt
