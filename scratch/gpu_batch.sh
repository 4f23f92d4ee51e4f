#!/bin/bash
# round 5, call 20: in-situ family calibration + pipelined FFN (v5): default vs forced families, same box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 5 900 python -m pytest tests/test_hip_round5.py -m gpu -q -x > $O/r05_c20_tests.log 2>&1; tail -2 $O/r05_c20_tests.log
run() { name=$1; shift
  env "$@" A2P_TUNE_VERBOSE=1 timeout -k 5 300 python bench.py --batch $B --no-cpu-baseline --no-parity --no-legs --steps 80 --warmup 10 > $O/r05_c20_$name.json 2> $O/r05_c20_$name.err
  grep "chain kernel family" $O/r05_c20_$name.err | head -2
  python - <<PY
import json
try:
    j=json.loads([l for l in open("$O/r05_c20_$name.json") if l.startswith("{")][-1])
    k=j["kernels"]; sub=k.get("_sub_classes",{})
    print("B=$B $name", j["value"], "hbm", j["box"].get("hbm_copy_gbs"), "family", j["roofline"].get("chain_family"), "chain", k["chain"]["ms_per_step"], {n:(v["avg_launch_us"]) for n,v in sub.items()})
except Exception as e:
    print("B=$B $name FAILED", e)
PY
}
for B in 8 32; do for rep in 1 2; do
  run gen1_$B A2P_CHAIN_V=1
  run tall_$B A2P_CHAIN_V=4
  run auto_$B A2P_CHAIN_V=0
done; done
