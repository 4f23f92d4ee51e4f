# correctness of the permuted-column chain kernels + quick perf (run under gpurun)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 120 scratch/chain_dbg | head -10
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
timeout 120 scratch/chain_bench 2>&1 | grep -E "M=|full|phases" | head -40
timeout 300 python bench.py --no-cpu-baseline --no-legs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:(v['avg_launch_us'], v.get('tflops')) for k,v in d['kernels'].items()})"
timeout 300 python bench.py --no-cpu-baseline --no-legs --batch 32 --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k:(v['avg_launch_us'], v.get('tflops')) for k,v in d['kernels'].items()})"
