R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "pose or body or shapes or identical or chain" 2>&1 | tail -3
j='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], {k:(v["launches_per_step"], v["avg_launch_us"]) for k,v in d["kernels"].items()})'
for rep in 1 2; do
A2P_CHAIN_NW=4 timeout 300 python bench.py --no-cpu-baseline --no-legs --no-parity --model pose --batch 16 2>/dev/null | python -c "$j" "pose NW4"
A2P_CHAIN_NW=8 timeout 300 python bench.py --no-cpu-baseline --no-legs --no-parity --model pose --batch 16 2>/dev/null | python -c "$j" "pose NW8 (MT5)"
done
A2P_CHAIN_NW=8 timeout 300 python bench.py --no-cpu-baseline --no-legs --no-parity --model pose --batch 8 2>/dev/null | python -c "$j" "pose B8 NW8"
A2P_CHAIN_NW=4 timeout 300 python bench.py --no-cpu-baseline --no-legs --no-parity --model pose --batch 8 2>/dev/null | python -c "$j" "pose B8 NW4"
