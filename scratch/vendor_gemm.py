"""Vendor-GEMM yardstick (scratch; never on the product path): what rocBLAS / hipBLASLt reach on THIS box for the decoder layer's
GEMM shapes as stand-alone launches ("also report against a measured hipBLASLt bf16 GEMM peak") -- the ceiling
of a non-fused design, before the HBM round trips of the [rows, d] activations between its launches are even counted.
    python scratch/vendor_gemm.py > profiles/r04_vendor_gemm.txt
y = x @ W^T with x [M, K], W [N, K] (torch.nn.functional.linear's layout), 16-bit operands, fp32 accumulate, 16-bit output.
"warm": the same operands every launch (L2/MALL-resident where they fit); "rotating": 8 operand sets in turn (what a step sees)."""
import os
import sys

import torch


def bench(M, N, K, dtype, rotating, iters=40):
    dev = torch.device("cuda:0")
    sets = 8 if rotating else 1
    xs = [torch.randn(M, K, device=dev, dtype=dtype) for _ in range(sets)]
    ws = [torch.randn(N, K, device=dev, dtype=dtype) * 0.05 for _ in range(sets)]
    out = torch.empty(M, N, device=dev, dtype=dtype)
    for i in range(5):
        torch.nn.functional.linear(xs[i % sets], ws[i % sets], out=None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        out = torch.nn.functional.linear(xs[i % sets], ws[i % sets])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    tf = 2.0 * M * N * K / us * 1e-6
    return us, tf


def main():
    assert torch.cuda.is_available()
    print(f"# torch {torch.__version__}, {torch.cuda.get_device_name(0)}, blas preference: {torch.backends.cuda.preferred_blas_library()}")
    print("# M x N x K, dtype: us per launch, TFLOP/s (fraction of the 2500 TFLOP/s dense 16-bit MFMA peak); HBM floor = (M*K + M*N) * 2 B at 8 TB/s")
    shapes = [(512, 512), (1024, 512), (512, 1024), (1536, 512)]
    for lib in ("default", "hipblaslt"):
        try:
            torch.backends.cuda.preferred_blas_library(lib if lib != "default" else "default")
        except Exception as e:   # noqa: BLE001
            print(f"# preferred_blas_library({lib}) not available: {e}")
            continue
        print(f"## blas library preference: {lib} -> {torch.backends.cuda.preferred_blas_library()}")
        for M in (38400, 9600):
            for (N, K) in shapes:
                for dtype in (torch.bfloat16, torch.float16):
                    row = []
                    for rot in (False, True):
                        us, tf = bench(M, N, K, dtype, rot)
                        row.append(f"{'rotating' if rot else 'warm'} {us:7.1f} us {tf:7.1f} TF/s ({tf / 2500:.3f})")
                    floor_us = (M * K + M * N) * 2 / 8e12 * 1e6
                    print(f"M={M:6d} N={N:5d} K={K:5d} {str(dtype).split('.')[-1]:9s}: " + " | ".join(row) + f" | HBM floor {floor_us:5.1f} us")
    # the fused chain's alternative: the FFN as two launches with the [M, 1024] hidden activation through HBM
    print("## FFN as two vendor launches (linear1 + GELU as a separate elementwise pass is NOT included): sum of the two GEMM times")
    for M in (38400, 9600):
        u1, _ = bench(M, 1024, 512, torch.float16, True)
        u2, _ = bench(M, 512, 1024, torch.float16, True)
        fl = 2.0 * M * 1024 * 512 * 2
        print(f"M={M}: {u1:.1f} + {u2:.1f} us = {fl / (u1 + u2) * 1e-6:.1f} TF/s ({fl / (u1 + u2) * 1e-6 / 2500:.3f})")


if __name__ == "__main__":
    sys.exit(main())
