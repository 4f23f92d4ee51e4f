R=$GRAFT_REPO_ROOT; cd $R
timeout 120 scratch/attn_bench 2>&1 | grep -E "nseq|abl= 0|abl=31|abl=24|abl= 4 "
timeout 400 python -m pytest tests -m gpu -q -x -k "attn or attention or parity or t600" 2>&1 | tail -3
j='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], {k:(v["launches_per_step"], v["avg_launch_us"]) for k,v in d["kernels"].items()})'
timeout 300 python bench.py --no-cpu-baseline --no-legs 2>/dev/null | python -c "$j" "B8"
timeout 300 python bench.py --no-cpu-baseline --no-legs --batch 32 --steps 10 2>/dev/null | python -c "$j" "B32"
