R=$GRAFT_REPO_ROOT; cd $R
j='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], {k:(v["launches_per_step"], v["avg_launch_us"]) for k,v in d["kernels"].items()})'
for cfg in "8 4" "8 3" "4 4" "4 3"; do set -- $cfg; A2P_CHAIN_NW=$1 A2P_CHAIN_MT=$2 timeout 300 python bench.py --no-cpu-baseline --no-legs --batch 32 --steps 10 2>/dev/null | python -c "$j" "B32 NW$1 MT$2"; done
timeout 300 python bench.py --no-cpu-baseline --no-legs --batch 32 --steps 10 2>/dev/null | python -c "$j" "B32 auto"
timeout 200 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "chain" 2>&1 | tail -2
