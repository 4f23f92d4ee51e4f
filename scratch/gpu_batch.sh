#!/bin/bash
# round 5, call 54: 48-row tall POST kernel with parked rows + 16-deep weight ring (scratch/ab/liba2p_park3_f16.so) vs the library: bit identity, then same-box A/B, families forced
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
A2P_LIB_F16=$R/scratch/ab/liba2p_park3_f16.so timeout -k 5 600 python -m pytest tests/test_hip_round5.py -m gpu -q -k "fp16" > $O/r05_c54_tests.log 2>&1; tail -2 $O/r05_c54_tests.log
for lib in lib park3 lib park3; do
  if [ $lib = park3 ]; then export A2P_LIB_F16=$R/scratch/ab/liba2p_park3_f16.so; else unset A2P_LIB_F16; fi
  A2P_CHAIN_V=4 timeout -k 5 300 python bench.py --batch 8 --no-cpu-baseline --no-parity --no-legs --steps 100 --warmup 10 > $O/r05_c54_b8_$lib.json 2> $O/r05_c54_b8_$lib.err
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/r05_c54_b8_$lib.json") if l.startswith("{")][-1])
    sub=j["kernels"]["_sub_classes"]; uid=[r for r in j["box"]["showhw"] if "Unique" in r]
    print("B=8 lib=$lib", j["value"], {n:v["avg_launch_us"] for n,v in sub.items()}, (j.get("under_load") or {}).get("sclk_mhz"), uid)
except Exception as e:
    print("B=8 lib=$lib FAILED", e); print(open("$O/r05_c54_b8_$lib.err").read()[-1200:])
PY
done 2>&1 | tee $O/r05_c54_ab.txt
