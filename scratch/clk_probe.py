"""In-step shader clock of the chain kernels (A2P_CHAIN_CLK=1): d(s_memtime) / d(s_memrealtime @ 100 MHz) per launch."""
import ctypes as C, os, sys
import numpy as np, torch
os.environ["A2P_CHAIN_CLK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from audio2photoreal_amd import _lib
dev = torch.device("cuda:0")
for fmt, B in (("face", 8), ("face", 32)):
    case = bench.Case(fmt, B, 600, "bf16", dev, list(range(B)))
    case.setup()
    for nw in ("4", "8"):
        os.environ["A2P_CHAIN_NW"] = nw
        with torch.no_grad():
            case.run_steps(6)
        torch.cuda.synchronize()
        a = np.zeros(64 * 32, np.uint64)
        _lib.check(case.model._lib().a2p_debug_read(case.model._ctx, b"clk", a.ctypes.data_as(C.c_void_p), a.nbytes), "clk")
        a = a.reshape(64, 8, 4).astype(np.float64)
        cyc, rt = a[:17, :, 2] - a[:17, :, 0], (a[:17, :, 3] - a[:17, :, 1]) * 10.0   # ns
        ok = rt > 0
        ghz = np.where(ok, cyc / np.maximum(rt, 1), 0)
        us = rt / 1e3
        print(f"{fmt} B={B} NW={nw}: per chain launch of one step (block 0): us = {np.round(us[:, 0], 1).tolist()}")
        print(f"   shader clock GHz (mean over blocks 0..7) = {np.round(ghz.mean(1), 2).tolist()}   overall {ghz[ok].mean():.3f} GHz")
    os.environ.pop("A2P_CHAIN_NW")
    case.model.release()
