"""Host enqueue time vs GPU time per denoise step (is the small configuration host-bound?)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
for fmt, B, T, prec in (("face", 1, 240, "fp16"), ("face", 1, 240, "fp32"), ("face", 8, 600, "fp16"), ("pose", 16, 600, "fp16")):
    case = bench.Case(fmt, B, T, prec, dev, list(range(B)), respacing="ddim10" if B == 1 else "", sampler="ddim" if B == 1 else "ddpm")
    case.setup()
    with torch.no_grad():
        case.run_steps(10)
        torch.cuda.synchronize()
        n = 100
        t0 = time.perf_counter()
        case.run_steps(n)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"{fmt} B={B} T={T} {prec}: host enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, wall {1e3 * (t2 - t0) / n:.3f} ms/step  ({n / (t2 - t0):.0f} steps/s)", flush=True)
    case.model.release()
