R=$GRAFT_REPO_ROOT; cd $R
j='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], {k:(v["avg_launch_us"], v.get("tflops")) for k,v in d["kernels"].items()})'
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "attention or forward or layer" 2>&1 | tail -2
for b in 8 32; do
A2P_ATTN_NO_REMAP=1 timeout 300 python bench.py --no-cpu-baseline --no-legs --batch $b --steps 10 2>/dev/null | python -c "$j" "B$b plain map, nt K/V     "
A2P_ATTN_NO_REMAP=1 A2P_KV_CACHED=1 timeout 300 python bench.py --no-cpu-baseline --no-legs --batch $b --steps 10 2>/dev/null | python -c "$j" "B$b plain map, cached K/V "
timeout 300 python bench.py --no-cpu-baseline --no-legs --batch $b --steps 10 2>/dev/null | python -c "$j" "B$b XCD remap, nt K/V     "
A2P_KV_CACHED=1 timeout 300 python bench.py --no-cpu-baseline --no-legs --batch $b --steps 10 2>/dev/null | python -c "$j" "B$b XCD remap, cached K/V "
done
timeout 300 python bench.py --no-cpu-baseline --no-legs --model pose --batch 16 --steps 10 2>/dev/null | python -c "$j" "pose16 remap nt"
A2P_KV_CACHED=1 timeout 300 python bench.py --no-cpu-baseline --no-legs --model pose --batch 16 --steps 10 2>/dev/null | python -c "$j" "pose16 remap cached"
A2P_ATTN_NO_REMAP=1 timeout 300 python bench.py --no-cpu-baseline --no-legs --model pose --batch 16 --steps 10 2>/dev/null | python -c "$j" "pose16 plain nt"
