R=$GRAFT_REPO_ROOT; cd $R
timeout 200 scratch/chain_bench 2>&1 | grep -E "M=|NW=8.*abl= 0 full|phases" 
timeout 600 python -m pytest tests -m gpu -q -x -k "chain or identical or hip_parity" 2>&1 | tail -3
j='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], {k:(v["launches_per_step"], v["avg_launch_us"]) for k,v in d["kernels"].items()})'
for nw in 4 8; do
A2P_CHAIN_NW=$nw timeout 300 python bench.py --no-cpu-baseline --no-legs --no-parity 2>/dev/null | python -c "$j" "B8 NW$nw"
done
