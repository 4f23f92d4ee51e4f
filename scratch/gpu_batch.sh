#!/bin/bash
# round 6, call 75: time-MLP table (three launches of the time path become a row lookup): full GPU suite; same-box A/B (A2P_TIME_TABLE=0) at B=8 / B=32, with and without the fused input kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 5 1500 python -X faulthandler -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/r06_gpu_tests_tt.log
for b in 8 32; do for v in "1000 0" "0 0" "1000 1" "1000 0" "0 0" "1000 1"; do
  set -- $v
  export A2P_TIME_TABLE=$1
  if [ $2 = 1 ]; then export A2P_NO_FUSED_IN=1; else unset A2P_NO_FUSED_IN; fi
  timeout -k 5 300 python bench.py --batch $b --no-cpu-baseline --no-parity --no-legs --steps 100 --warmup 8 > $O/r06_c75.json 2>/dev/null
  python - <<PY
import json
j=json.loads([l for l in open("$O/r06_c75.json") if l.startswith("{")][-1])
k=j["kernels"]; sub=k["_sub_classes"]
print("B=$b time_table=$1 no_fused_in=$2", j["value"], "steps/s", {a:v["avg_launch_us"] for a,v in sub.items()}, "family", j["roofline"].get("chain_family"))
PY
done; done | tee $O/r06_time_table_ab.txt
