"""Side-stream (A2P_SIDE_STREAM=1) race screen: N guided forwards at bench size must be bit-identical to the single-stream result."""
import os, sys, hashlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for prec in ("fp16", "bf16"):
    case = bench.Case("face", 8, 600, prec, dev, list(range(8)))
    case.setup()
    t = torch.tensor([999, 750, 500, 250, 100, 10, 1, 0], device=dev)
    os.environ.pop("A2P_SIDE_STREAM", None)
    with torch.no_grad():
        ref = case.cfg(case.x, t, case.y).clone()
        os.environ["A2P_SIDE_STREAM"] = "1"
        bad = 0
        for i in range(N):
            out = case.cfg(case.x, case.steps_idx[(i * 37) % 1000] if i % 2 else t, case.y)
            if i % 2 == 0 and not torch.equal(out, ref):
                bad += 1
        # interleave with the sampler step (x changes every step) and compare whole trajectories
        traj = []
        for mode in ("0", "1"):
            if mode == "1": os.environ["A2P_SIDE_STREAM"] = "1"
            else: os.environ.pop("A2P_SIDE_STREAM", None)
            case.state = {"x": case.x, "i": case.n_chain - 1}
            case.gen.manual_seed(99)
            case.run_steps(60)
            traj.append(case.state["x"].clone())
    print(f"{prec}: {bad} of {N // 2} side-stream forwards differ; 60-step trajectories equal: {torch.equal(traj[0], traj[1])}", flush=True)
    case.model.release()
