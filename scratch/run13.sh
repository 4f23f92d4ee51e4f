R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for cfg in "1 240" "4 600" "8 600"; do set -- $cfg
for v in "A2P_X=1" "A2P_NO_CHAIN=1"; do
env $v timeout 300 python bench.py --steps 50 --warmup 5 --batch $1 --frames $2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$1 T=$2 $v', d['value'], d['ms_per_step'])"
done; done
