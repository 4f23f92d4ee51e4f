R=$GRAFT_REPO_ROOT; cd $R
j='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], {k:(v["launches_per_step"], v["avg_launch_us"]) for k,v in d["kernels"].items()})'
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 300 python scratch/side_stress.py 200
A2P_NO_SHARED_HALF=1 A2P_NO_SIDE_STREAM=1 timeout 300 python bench.py --no-cpu-baseline --no-legs 2>/dev/null | python -c "$j" "B8 base            "
A2P_NO_SHARED_HALF=1 timeout 300 python bench.py --no-cpu-baseline --no-legs 2>/dev/null | python -c "$j" "B8 +side stream     "
A2P_NO_SIDE_STREAM=1 timeout 300 python bench.py --no-cpu-baseline --no-legs 2>/dev/null | python -c "$j" "B8 +shared half     "
timeout 300 python bench.py --no-cpu-baseline --no-legs 2>/dev/null | python -c "$j" "B8 both             "
timeout 300 python bench.py --no-cpu-baseline --no-legs --batch 32 --steps 10 2>/dev/null | python -c "$j" "B32 both"
timeout 300 python bench.py --no-cpu-baseline --no-legs --model pose --batch 16 2>/dev/null | python -c "$j" "pose16 both"
