R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
j='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], {k:(v["launches_per_step"], v["avg_launch_us"]) for k,v in d["kernels"].items()})'
for rep in 1 2; do
timeout 300 python bench.py --batch 1 --frames 240 --steps 50 --no-cpu-baseline --no-legs --no-parity 2>/dev/null | python -c "$j" "cfg0 ring4"
A2P_GEMM_RING2=1 timeout 300 python bench.py --batch 1 --frames 240 --steps 50 --no-cpu-baseline --no-legs --no-parity 2>/dev/null | python -c "$j" "cfg0 ring2"
done
timeout 300 python bench.py --no-cpu-baseline --no-legs --no-parity 2>/dev/null | python -c "$j" "B8 ring4"
A2P_GEMM_RING2=1 timeout 300 python bench.py --no-cpu-baseline --no-legs --no-parity 2>/dev/null | python -c "$j" "B8 ring2"
