R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -E "edge|passed|failed|Error|error|assert" | tail -20
timeout 300 python bench.py --model pose --batch 16 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pose b16', d['value'], d['ms_per_step']); [print(k, v['ms_per_step'], v['avg_launch_us'], v.get('tflops')) for k,v in d['kernels'].items()]"
A2P_NO_CHAIN=1 timeout 300 python bench.py --model pose --batch 16 --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pose b16 per-op', d['value'], d['ms_per_step'])"
