#!/bin/bash
# round 6, call 65: same-box phase stamps, round-5 feed-forward block (barriers) vs the pipelined one; chain launch times at B=4 (100 workgroups) vs B=8 (200)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
for lib in stamps stamps2; do
  export A2P_LIB_F16=$R/scratch/ab/liba2p_${lib}_f16.so
  for b in 8 32; do PP_BATCH=$b timeout -k 5 300 python scratch/phase_probe4.py 2>/dev/null | grep "gen 4" | sed "s/^/$lib /"; done
done | tee $O/r06_ffn_pipe_stamps_ab.txt
unset A2P_LIB_F16
for b in 4 8; do
  timeout -k 5 300 python bench.py --batch $b --no-cpu-baseline --no-parity --no-legs --steps 60 --warmup 8 > $O/r06_c65.json 2>/dev/null
  python - <<PY
import json
j=json.loads([l for l in open("$O/r06_c65.json") if l.startswith("{")][-1])
k=j["kernels"]; sub=k["_sub_classes"]
print("B=$b", j["value"], "steps/s", {a:v["avg_launch_us"] for a,v in sub.items()}, {x:k[x]["avg_launch_us"] for x in ("attn_self","attn_cross")}, "family", j["roofline"].get("chain_family"))
PY
done | tee $O/r06_chain_b4_vs_b8.txt
