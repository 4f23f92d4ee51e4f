set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"
./scratch/gemm_bench > gpurun_out/gemm_bench.log 2>&1
python scratch/ref_lib.py > gpurun_out/ref_lib.log 2>&1
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_b8.json 2> gpurun_out/bench_b8.err
python bench.py --steps 10 --warmup 3 --batch 32 --no-cpu-baseline > gpurun_out/bench_b32.json 2> gpurun_out/bench_b32.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v2 -o v2 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_v2.log 2>&1
ls -R $R/gpurun_out/prof_v2 | head
