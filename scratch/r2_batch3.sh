R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
j='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], {k:(v["avg_launch_us"], v.get("tflops")) for k,v in d["kernels"].items()})'
A2P_TUNE_VERBOSE=1 timeout 300 python bench.py --no-cpu-baseline --no-legs 2>&1 | grep -E "a2p\]|metric" | cut -c1-200 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): 
        pass
    print(l[:200].rstrip())"
timeout 300 python bench.py --no-cpu-baseline --no-legs 2>/dev/null | python -c "$j" "B8 auto"
A2P_CHAIN_NW=4 timeout 300 python bench.py --no-cpu-baseline --no-legs 2>/dev/null | python -c "$j" "B8 NW4"
A2P_CHAIN_NW=8 timeout 300 python bench.py --no-cpu-baseline --no-legs 2>/dev/null | python -c "$j" "B8 NW8"
for cfg in "4 4" "4 3" "8 3"; do set -- $cfg; A2P_CHAIN_NW=$1 A2P_CHAIN_MT=$2 timeout 300 python bench.py --no-cpu-baseline --no-legs --batch 32 --steps 10 2>/dev/null | python -c "$j" "B32 NW$1 MT$2"; done
timeout 120 scratch/chain_bench 2>&1 | grep -E "M=|full|phases" | tail -14
