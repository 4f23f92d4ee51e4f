R=$GRAFT_REPO_ROOT
cd $R
bash scratch/run_pmc.sh r01_v3
cp gpurun_out/pmc_traffic_r01_v3.json profiles/pmc_traffic.json
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_v3.json 2> gpurun_out/bench_v3.err; tail -c 2500 gpurun_out/bench_v3.json
