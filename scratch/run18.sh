R=$GRAFT_REPO_ROOT
cd $R
echo default; ./scratch/chain_bench 2>&1 | grep -E "abl= 0 " | tail -3
echo nt-stores; ./scratch/chain_bench_ntst 2>&1 | grep -E "abl= 0 " | tail -3
for i in 1 2; do
for v in "A2P_X=1" "A2P_LIB=$R/scratch/liba2p_ntst.so"; do
env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$i $v', d['value'], d['ms_per_step'])"
done; done
