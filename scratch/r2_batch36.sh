R=$GRAFT_REPO_ROOT; cd $R
j='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], {k:(v["launches_per_step"], v["avg_launch_us"]) for k,v in d["kernels"].items()})'
for rep in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-legs --no-parity 2>/dev/null | python -c "$j" "B8 nt"
A2P_KV_CACHED=1 timeout 300 python bench.py --no-cpu-baseline --no-legs --no-parity 2>/dev/null | python -c "$j" "B8 cached"
done
timeout 300 python bench.py --no-cpu-baseline --no-legs --no-parity --batch 32 --steps 10 2>/dev/null | python -c "$j" "B32 nt"
A2P_KV_CACHED=1 timeout 300 python bench.py --no-cpu-baseline --no-legs --no-parity --batch 32 --steps 10 2>/dev/null | python -c "$j" "B32 cached"
A2P_NO_SIDE_STREAM=1 timeout 300 python bench.py --no-cpu-baseline --no-legs --no-parity 2>/dev/null | python -c "$j" "B8 no-side"
A2P_NO_SHARED_HALF=1 timeout 300 python bench.py --no-cpu-baseline --no-legs --no-parity 2>/dev/null | python -c "$j" "B8 no-shared-half"
