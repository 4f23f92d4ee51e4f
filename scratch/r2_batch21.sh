R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
cp gpurun_out/parity_tests.json gpurun_out/r02_parity_tests_full.json 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -2
timeout 600 python bench.py --write-parity gpurun_out/r02_parity.json > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.json
bash scratch/run_pmc.sh r02
