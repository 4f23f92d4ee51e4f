R=$GRAFT_REPO_ROOT
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
bash scratch/run_pmc.sh r01_v4 > /dev/null 2>&1
cp gpurun_out/pmc_traffic_r01_v4.json profiles/pmc_traffic.json
python bench.py > gpurun_out/bench_v4.json 2> gpurun_out/bench_v4.err; tail -c 2600 gpurun_out/bench_v4.json
python bench.py --batch 32 --steps 10 --no-cpu-baseline > gpurun_out/bench_v4_b32.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/bench_v4_b32.json').read().strip().splitlines()[-1]); print('b32', d['value'], d['ms_per_step'], d['decoder_mfma_frac']); [print(k, v['ms_per_step'], v['avg_launch_us'], v.get('tflops')) for k,v in d['kernels'].items()]"
./scratch/attn_bench > gpurun_out/attn_bench_v4.log 2>&1
./scratch/chain_bench > gpurun_out/chain_bench_v4.log 2>&1
