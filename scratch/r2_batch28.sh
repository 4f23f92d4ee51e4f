R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python bench.py --pipeline 2>/dev/null | tail -1 | cut -c1-900
timeout 600 python bench.py --pipeline --batch 32 2>/dev/null | tail -1 | cut -c1-900
export A2P_BENCH_SHARE_GPU=1 A2P_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --no-cpu-baseline --no-legs --no-parity 2>/dev/null | tail -1 | cut -c1-700
