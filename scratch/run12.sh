R=$GRAFT_REPO_ROOT
cd $R
./scratch/membench
./scratch/chain_bench 2>&1 | grep -E "MT=3 mode=[12] abl= 0" | head -2
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench b8', d['value'], d['ms_per_step'], d['kernels']['chain']['avg_launch_us'], d['kernels']['attn_cross']['avg_launch_us'])"
rocm-smi --showclocks --showpower --showmemuse --showperflevel 2>/dev/null | grep -E "clk|Power|perf|Perf" | head
