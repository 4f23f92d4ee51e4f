"""In-step phase times of one chain launch (diagnostic build: -DA2P_STAMPS, A2P_LIB_F16=scratch/ab/liba2p_stamps_f16.so):
launch A2P_STAMP_LAUNCH (default 4 = layer-1 POST) of every forward writes 100 MHz stamps at its phase boundaries."""
import ctypes as C, os, sys
import numpy as np, torch
os.environ["A2P_CHAIN_CLK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from audio2photoreal_amd import _lib
dev = torch.device("cuda:0")
G1 = ["prologue", "out_proj", "film_res", "ln_stats", "ln_write", "ffn", "film_res", "(store)", "ln+rope", "qk_gemm", "ln+v_gemm", "drain"]
G2 = {1: "prologue", 2: "out_proj", 3: "film_res", 4: "ln+park", 5: "ffn", 6: "film_res", 9: "ln+rope+store", 10: "qk_gemm", 11: "reload+ln", 12: "v_gemm"}
B = int(os.environ.get("PP_BATCH", "8"))
case = bench.Case("face", B, 600, "fp16", dev, list(range(B)))
case.setup()
for ver, nw in (("1", "8"), ("1", "4")):
    os.environ["A2P_CHAIN_V"], os.environ["A2P_CHAIN_NW"] = ver, nw
    with torch.no_grad():
        case.run_steps(6)
    torch.cuda.synchronize()
    a = np.zeros(64 * 32 + 128, np.uint64)
    _lib.check(case.model._lib().a2p_debug_read(case.model._ctx, b"clk", a.ctypes.data_as(C.c_void_p), a.nbytes), "clk")
    st = a[64 * 32: 64 * 32 + 64].astype(np.float64) * 0.01     # us
    for blk in (0, 1):
        h = st[blk * 32: blk * 32 + 13]
        if ver == "1":
            parts = [f"{G1[i - 1]}={h[i] - h[i - 1]:.2f}" for i in range(1, 13) if h[i] > 0 and h[i - 1] > 0]
        else:
            parts, prev = [], 0
            for i, n in G2.items():
                if h[i] > 0:
                    parts.append(f"{n}={h[i] - h[prev]:.2f}")
                    prev = i
        print(f"gen {ver} NW={nw} block {'0' if blk == 0 else '101'}: total={h[12] - h[0]:.2f} us | " + " ".join(parts))
        if ver == "1":   # gemm_store's per-tile stamps (slots 13..30): tile GEMM done, tile stored, ...
            e = st[blk * 32: blk * 32 + 32]
            seq = [e[9]] + [x for x in e[13:31] if x > 0]
            print("      [Q|K] phase from its start, per tile (gemm, epilogue) us: " +
                  " ".join(f"({seq[i + 1] - seq[i]:.2f},{seq[i + 2] - seq[i + 1]:.2f})" for i in range(0, len(seq) - 2, 2)))
