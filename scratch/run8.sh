R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for v in "" "A2P_NO_SIDE_STREAM=1"; do
env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench b8 $v', d['value'], d['ms_per_step'])"
done
env timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench b8 again', d['value'], d['ms_per_step'])"
