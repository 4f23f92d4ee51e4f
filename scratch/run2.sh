R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests -m gpu -x -q -k "chain or decoder_layer or forward_vs" 2>&1 | tail -25 > gpurun_out/gpu_tests_chain.log; cat gpurun_out/gpu_tests_chain.log
for mt in 2 3 4; do
A2P_CHAIN_MT=$mt timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_chain_mt$mt.json 2> gpurun_out/bench_chain_mt$mt.err
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_chain_mt$mt.json').read().strip().splitlines()[-1]); print($mt, d['value'], d['ms_per_step'], d['kernels'].get('chain'))
PY
done
timeout 300 python bench.py --steps 10 --warmup 3 --batch 32 --no-cpu-baseline > gpurun_out/bench_chain_b32.json 2> gpurun_out/bench_chain_b32.err
python -c "
import json
d=json.loads(open('gpurun_out/bench_chain_b32.json').read().strip().splitlines()[-1]); print('b32', d['value'], d['ms_per_step'], d['kernels'].get('chain'))"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc1 -o p -- $B > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc2 -o p -- $B > $R/gpurun_out/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc3 -o p -- $B > $R/gpurun_out/pmc3.log 2>&1
cd $R
for i in 1 2 3; do f=$(ls gpurun_out/pmc$i/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python scratch/pmc_summary.py $f > gpurun_out/pmc${i}_summary.txt; grep chain gpurun_out/pmc${i}_summary.txt; rm -f gpurun_out/pmc$i/*counter_collection.csv gpurun_out/pmc$i/*kernel_trace.csv; done
