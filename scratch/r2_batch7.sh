R=$GRAFT_REPO_ROOT; cd $R
j='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(sys.argv[1], "ranks", d["n_gpus"], "value", d["value"], "sample_steps/s", d["sample_steps_per_sec"]*d["n_gpus"] if False else round(d["value"]*d["config"]["global_batch"]/d["n_gpus"],1), "ms/step", d["ms_per_step"])'
run() { A2P_BENCH_SHARE_GPU=1 A2P_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $((29600+$1+$2)) bench.py --gpus $1 --batch $2 --steps 10 --warmup 2 --no-kernel-timing 2>/dev/null | python -c "$j" "B=$2/rank"; }
timeout 300 python bench.py --no-cpu-baseline --no-legs --no-kernel-timing 2>/dev/null | python -c "$j" "single B=8"
run 2 4
run 2 8
run 4 2
run 4 4
timeout 300 python bench.py --no-cpu-baseline --no-legs --no-kernel-timing --batch 32 --steps 10 2>/dev/null | python -c "$j" "single B=32"
run 2 16
run 4 8
