R=$GRAFT_REPO_ROOT; cd $R
j='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], {k:(v["launches_per_step"], v["avg_launch_us"]) for k,v in d["kernels"].items()})'
for nw in 4 8; do for w in 0 1; do
if [ $w = 1 ]; then export A2P_CHAIN_WARM=1; else unset A2P_CHAIN_WARM; fi
A2P_CHAIN_NW=$nw timeout 300 python bench.py --no-cpu-baseline --no-legs --no-parity 2>/dev/null | python -c "$j" "B8 NW$nw warm=$w"
done; done
