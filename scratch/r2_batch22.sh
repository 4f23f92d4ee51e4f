R=$GRAFT_REPO_ROOT; cd $R
j='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], {k:(v["launches_per_step"], v["avg_launch_us"]) for k,v in d["kernels"].items()})'
for rep in 1 2; do for prec in fp16 bf16; do for nw in 4 8; do
A2P_CHAIN_NW=$nw timeout 300 python bench.py --precision $prec --no-cpu-baseline --no-legs --no-parity 2>/dev/null | python -c "$j" "B8 $prec NW$nw"
done; done; done
A2P_TUNE_VERBOSE=1 timeout 300 python bench.py --precision fp16 --no-cpu-baseline --no-legs --no-parity 2>&1 | grep "a2p\]" | head -3
A2P_TUNE_VERBOSE=1 timeout 300 python bench.py --precision bf16 --no-cpu-baseline --no-legs --no-parity 2>&1 | grep "a2p\]" | head -3
