"""Known-good reference rates on the same box (scratch; not product): what the vendor libraries reach on
the denoiser's GEMM / attention shapes (hipBLASLt via torch.matmul, flash attention via torch SDPA)."""
import torch
import torch.nn.functional as F


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


dev = torch.device("cuda")
for (M, N, K) in [(9600, 512, 512), (9600, 1024, 512), (9600, 512, 1024), (38400, 512, 512), (38400, 1024, 512), (38400, 512, 1024),
                  (4096, 4096, 4096)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    us = timeit(lambda: F.linear(a, w))
    print(f"hipblaslt linear M={M} N={N} K={K}: {us:8.1f} us  {2.0 * M * N * K / us * 1e-6:8.1f} TF")
for (nseq, T, S) in [(16, 600, 600), (16, 600, 2000), (64, 600, 600), (64, 600, 2000)]:
    q = torch.randn(nseq, 8, T, 64, device=dev, dtype=torch.bfloat16)
    k = torch.randn(nseq, 8, S, 64, device=dev, dtype=torch.bfloat16)
    v = torch.randn(nseq, 8, S, 64, device=dev, dtype=torch.bfloat16)
    try:
        us = timeit(lambda: F.scaled_dot_product_attention(q, k, v))
        print(f"torch sdpa nseq={nseq} T={T} S={S}: {us:8.1f} us  {4.0 * nseq * 8 * T * S * 64 / us * 1e-6:8.1f} TF")
    except Exception as e:  # noqa: BLE001
        print("sdpa failed", e)
# fp32 elementwise RMW of the residual stream (the FiLM epilogue's traffic) as a bandwidth yardstick
x = torch.randn(9600, 512, device=dev)
y = torch.randn(9600, 512, device=dev)
us = timeit(lambda: x.add_(y))
print(f"x.add_(y) 9600x512 fp32: {us:.1f} us  ({3 * x.numel() * 4 / us * 1e-3:.0f} GB/s)")
