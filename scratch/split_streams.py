"""Would two half-batches on two HIP streams beat one batch?  (scratch; not product)
At B=8 the chain kernels run one 48-row panel per CU on 200 of 256 CUs and the attention launches are 2.5 workgroups per CU; kernel
boundaries drain and refill the chip 33 times per step.  Two independent half-batches (two contexts, B=4 each, the kernel family of
the whole batch through the batch hint) on two streams let one half's attention fill the CUs the other half's chain panels leave.
    python scratch/split_streams.py [B] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
fmt = sys.argv[3] if len(sys.argv) > 3 else "face"
dev = torch.device("cuda:0")
whole = bench.Case(fmt, B, 600, "fp16", dev, list(range(B)))
whole.setup()
halves = [bench.Case(fmt, B // 2, 600, "fp16", dev, list(range(h * B // 2, (h + 1) * B // 2))) for h in range(2)]
for c in halves:
    c.model.global_batch_hint = B
    c.setup()
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]

def run_whole(n):
    whole.run_steps(n)

def run_split(n):
    for _ in range(n):
        for c, s in zip(halves, streams):
            with torch.cuda.stream(s):
                c.run_steps(1)

def timed(f, n):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(n); torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

with torch.no_grad():
    run_whole(10); run_split(10)
    for rep in range(3):
        a = timed(run_whole, K); b = timed(run_split, K)
        print(f"{fmt} B={B}: one batch {a:.3f} ms/step ({1e3 / a:.1f} steps/s)   two half-batches on two streams {b:.3f} ms/step ({1e3 / b:.1f} steps/s)   {a / b:.3f}x", flush=True)
    # host-only enqueue time of the split schedule (is the host the limit?)
    t0 = time.perf_counter(); run_split(K); t_enq = (time.perf_counter() - t0) / K * 1e3; torch.cuda.synchronize()
    print(f"host enqueue time of the split schedule: {t_enq:.3f} ms/step")
    # same samples?
    whole.state = {"x": whole.x, "i": whole.n_chain - 1}
    for c in halves: c.state = {"x": c.x, "i": c.n_chain - 1}
    whole.gen.manual_seed(1); [c.gen.manual_seed(1) for c in halves]
