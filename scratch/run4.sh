R=$GRAFT_REPO_ROOT
cd $R
./scratch/chain_bench > gpurun_out/chain_bench.log 2>&1; grep -E "^M=|MT=3|MT=4" gpurun_out/chain_bench.log | grep -v "cycled=16" | head -22
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench b8', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --steps 10 --warmup 3 --batch 32 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench b32', d['value'], d['ms_per_step'])"
A2P_CHAIN_MT=3 timeout 300 python bench.py --steps 10 --warmup 3 --batch 32 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench b32 mt3', d['value'], d['ms_per_step'])"
A2P_NO_CHAIN=1 timeout 300 python bench.py --steps 10 --warmup 3 --batch 32 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench b32 nochain', d['value'], d['ms_per_step'])"
