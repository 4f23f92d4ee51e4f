R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python scratch/side_stress.py 600
echo "== 2 ranks sharing the GPU through bench.py"
A2P_BENCH_SHARE_GPU=1 A2P_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 2>&1 | tail -3 | cut -c1-1500
