R=$GRAFT_REPO_ROOT
cd $R
./scratch/chain_bench 2>&1 | grep -E "^M=|MT=" | grep -v "cycled=16" | head -25
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench b8', d['value'], d['ms_per_step']); [print(k, v['ms_per_step'], v['avg_launch_us'], v.get('tflops')) for k,v in d['kernels'].items()]"; done
timeout 300 python bench.py --steps 10 --warmup 3 --batch 32 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench b32', d['value'], d['ms_per_step'])"
