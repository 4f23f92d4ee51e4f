# Fingerprint of the GPU box this gpurun call landed on: the bench (with the shape the library picks) next to both forced
# workgroup shapes and the chain microbench, to characterise the "slow" group of boxes.
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; O=$R/gpurun_out/box_probe_$(date +%H%M%S).txt
j='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], {k: v["avg_launch_us"] for k, v in d.get("kernels", {}).items()})'
{
  echo "== $(date)"
  A2P_TUNE_VERBOSE=1 python $R/bench.py --no-cpu-baseline 2>&1 | grep -E "a2p|metric" | cut -c1-120
  A2P_CHAIN_NW=4 python $R/bench.py --no-cpu-baseline 2>/dev/null | python -c "$j" "NW=4"
  A2P_CHAIN_NW=8 python $R/bench.py --no-cpu-baseline 2>/dev/null | python -c "$j" "NW=8"
  timeout 60 $R/scratch/chain_bench 2>&1 | grep -E "M=|abl= 0 full"
} > $O 2>&1
cat $O
