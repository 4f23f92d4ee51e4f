R=$GRAFT_REPO_ROOT
cd $R
./scratch/chain_bench 2>&1 | grep -E "^M=|abl= [01] "
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['kernels']['chain']['avg_launch_us'], d['kernels']['attn_cross']['avg_launch_us'])"; done
timeout 300 python bench.py --steps 10 --warmup 3 --batch 32 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b32', d['value'], d['ms_per_step'])"
