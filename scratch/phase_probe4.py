"""In-step phase times of one tall-chain launch (kernels_chain4.h; diagnostic build -DA2P_STAMPS into scratch/ab/, A2P_LIB_F16=...):
launch A2P_STAMP_LAUNCH of every forward (default 4 = layer-1 POST, 2 = layer-0 MID ... see a2p_lib_run.h chain_base) writes 100 MHz
stamps at its phase boundaries.  PP_BATCH = samples (x2 guidance)."""
import ctypes as C, os, sys
import numpy as np, torch
os.environ["A2P_CHAIN_CLK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from audio2photoreal_amd import _lib
dev = torch.device("cuda:0")
G1 = ["prologue", "out_proj", "film_res", "ln_stats", "ln_write", "ffn", "film_res", "(store)", "ln+rope", "qk_gemm", "ln+v_gemm", "drain"]
G4P = {1: "prologue", 2: "out_proj", 3: "film_res", 4: "ln_stats", 5: "ln_write+park", 6: "ffn", 7: "film_res(reload)", 8: "ln_stats+ln_rope", 9: "store_x", 10: "qk_gemm", 11: "reload+ln", 12: "v_gemm"}
G4M = {1: "prologue", 2: "out_proj", 3: "film_res", 4: "ln_stats", 5: "ln_rope", 6: "q_gemm", 7: "store_x"}
B = int(os.environ.get("PP_BATCH", "32"))
case = bench.Case("face", B, 600, "fp16", dev, list(range(B)))
case.setup()
for ver in ("1", "4"):
    os.environ["A2P_CHAIN_V"] = ver
    with torch.no_grad():
        case.run_steps(6)
    torch.cuda.synchronize()
    a = np.zeros(64 * 32 + 128, np.uint64)
    _lib.check(case.model._lib().a2p_debug_read(case.model._ctx, b"clk", a.ctypes.data_as(C.c_void_p), a.nbytes), "clk")
    raw = a[64 * 32: 64 * 32 + 64].astype(np.float64)
    st = raw * 0.01     # us
    for blk in (0, 1):
        h = st[blk * 32: blk * 32 + 32]
        names = G4M if (ver == "4" and os.environ.get("A2P_STAMP_LAUNCH") in ("1", "2", "3")) else G4P
        if ver == "1":
            parts = [f"{G1[i - 1]}={h[i] - h[i - 1]:.2f}" for i in range(1, 13) if h[i] > 0 and h[i - 1] > 0]
            tot = h[12] - h[0]
        else:
            parts, prev = [], 0
            for i, n in names.items():
                if h[i] > 0:
                    parts.append(f"{n}={h[i] - h[prev]:.2f}")
                    prev = i
            tot = h[prev] - h[0]
            if names is G4P and h[13] > 0:   # feed-forward block, per hidden chunk: linear1 | GELU + barriers | linear2 (wave 0's view)
                ff, prev = [], 5
                for c in range(6):
                    i = 13 + 3 * c
                    if i + 2 < 30 and h[i + 2] > 0:
                        ff.append(f"[{h[i] - h[prev]:.2f} {h[i + 1] - h[i]:.2f} {h[i + 2] - h[i + 1]:.2f}]")
                        prev = i + 2
                parts.append("ffn chunks (lin1 gelu+bar lin2): " + " ".join(ff))
        if ver == "4" and raw[blk * 32 + 31] > raw[blk * 32 + 30] > 0 and tot > 0:
            parts.append(f"| {raw[blk * 32 + 31] - raw[blk * 32 + 30]:.0f} shader cycles = {(raw[blk * 32 + 31] - raw[blk * 32 + 30]) / tot * 1e-3:.3f} GHz effective")
        print(f"B={B} gen {ver} block {'0' if blk == 0 else '101'}: total={tot:.2f} us | " + " ".join(parts), flush=True)
