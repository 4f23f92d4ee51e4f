R=$GRAFT_REPO_ROOT; cd $R
timeout 200 scratch/chain_bench 2>&1 | grep -E "M=|NW=8.*abl= 0 full|NW=8.*no stores$|phases" 
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
j='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], {k:(v["launches_per_step"], v["avg_launch_us"]) for k,v in d["kernels"].items()})'
for nw in 4 8; do
A2P_CHAIN_NW=$nw timeout 300 python bench.py --no-cpu-baseline --no-legs --no-parity 2>/dev/null | python -c "$j" "B8 NW$nw"
done
A2P_CHAIN_NW=8 timeout 300 python bench.py --no-cpu-baseline --no-legs --no-parity --batch 32 --steps 10 2>/dev/null | python -c "$j" "B32 NW8"
A2P_CHAIN_NW=8 timeout 300 python bench.py --no-cpu-baseline --no-legs --no-parity --model pose --batch 16 2>/dev/null | python -c "$j" "pose NW8"
A2P_CHAIN_NW=4 timeout 300 python bench.py --no-cpu-baseline --no-legs --no-parity --model pose --batch 16 2>/dev/null | python -c "$j" "pose NW4"
