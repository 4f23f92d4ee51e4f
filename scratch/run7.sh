R=$GRAFT_REPO_ROOT
cd $R
./scratch/attn_bench 2>&1 | grep -E "nseq|abl= 0|abl=31|abl=24|abl= 3 "
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench b8', d['value'], d['ms_per_step']); [print(k, v['ms_per_step'], v['avg_launch_us'], v.get('tflops')) for k,v in d['kernels'].items()]"
done
