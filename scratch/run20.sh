R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
bash scratch/run_pmc.sh r01_v5 > /dev/null 2>&1
cp gpurun_out/pmc_traffic_r01_v5.json profiles/pmc_traffic.json
python bench.py > gpurun_out/bench_v5.json 2> gpurun_out/bench_v5.err; python -c "
import json; d=json.loads(open('gpurun_out/bench_v5.json').read().strip().splitlines()[-1]); print('b8', d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value']); [print(k, v['ms_per_step'], v['avg_launch_us'], v.get('tflops')) for k,v in d['kernels'].items()]"
python bench.py --batch 32 --steps 10 --no-cpu-baseline > gpurun_out/bench_v5_b32.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/bench_v5_b32.json').read().strip().splitlines()[-1]); print('b32', d['value'], d['ms_per_step'], d['decoder_mfma_frac'])"
