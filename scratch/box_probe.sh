# Fingerprint of the GPU box this gpurun call landed on: the bench's per-kernel averages next to the chain microbench
# (warm / L2+MALL-thrashed), to find out what the "slow" boxes of DESIGN.md section 6 have in common.
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; O=$R/gpurun_out/box_probe_$(date +%H%M%S).txt
{
  echo "== $(date) $(hostname)"; /opt/rocm/bin/rocm-smi --showclocks --showmemuse --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power|GPU\[0\]" | head -12
  python $R/bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], {k:(v['avg_launch_us']) for k,v in d['kernels'].items()})"
  timeout 60 $R/scratch/chain_bench 2>&1 | grep -E "M=|abl= 0 full|no dma|block   0"
  A2P_CHAIN_NW=8 python $R/bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench NW8', d['value'], d['ms_per_step'])"
  A2P_KV_CACHED=1 python $R/bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench cached-KV', d['value'], d['ms_per_step'])"
} > $O 2>&1
cat $O
