R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_hip_round2.py -m gpu -q -x -k "mixed_panel" 2>&1 | tail -3
j='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], {k:(v["launches_per_step"], v["avg_launch_us"]) for k,v in d["kernels"].items()})'
for nw in 4 8; do for mix in 1 0; do
if [ $mix = 0 ]; then export A2P_CHAIN_NO_MIX=1; else unset A2P_CHAIN_NO_MIX; fi
A2P_CHAIN_NW=$nw timeout 300 python bench.py --no-cpu-baseline --no-legs --no-parity --batch 32 --steps 10 2>/dev/null | python -c "$j" "B32 NW$nw mix=$mix"
done; done
unset A2P_CHAIN_NO_MIX
timeout 300 python bench.py --no-cpu-baseline --no-legs --no-parity --batch 32 --steps 10 2>/dev/null | python -c "$j" "B32 auto"
