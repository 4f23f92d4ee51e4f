R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
for v in "A2P_X=1" "A2P_NO_ARENA=1"; do
env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$i $v', d['value'], d['ms_per_step'], d['kernels']['chain']['avg_launch_us'], d['kernels']['attn_cross']['avg_launch_us'])"
done; done
