"""ct = cond_projection(pre_audio) of the guide's hoisted conditioning differs between the first process on a fresh box and later
ones while pre_audio does not (gpurun_out/c8_determinism.txt).  Compare both against a float64 torch product."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from audio2photoreal_amd import _lib  # noqa: E402
import argparse  # noqa: E402


def read(g, name, shape):
    host = np.empty(shape, np.float32)
    _lib.check(_lib.load().a2p_guide_debug_read(g._ctx, name.encode(), host.ctypes.data_as(C.c_void_p), host.nbytes), name)
    return torch.from_numpy(host)


def main():
    a = argparse.Namespace(frames=240, respacing="ddim5", precision="fp16", batch=2)
    dev = torch.device("cuda:0")
    subj = bench.PipelineSubject(a, dev, 0, [0, 1])
    feats = subj.models["pose"][1].model.audio_frontend.encode_audio(subj.audio)
    g = subj.guide
    B, S = feats.shape[0], feats.shape[1]
    for it in range(2):
        g.invalidate_cond()
        g._prepare(feats, 0.0)
        pre = read(g, "pre_audio", (B * S, 1024))
        ct = read(g, "ct", (B * S, 64))
        W = g.cond_projection.weight.detach().cpu().double()
        b = g.cond_projection.bias.detach().cpu().double()
        want = pre.double() @ W.T + b
        err = (ct.double() - want).abs()
        rows_bad = (err.max(dim=1).values > 1e-3).nonzero().flatten()
        print(f"prepare {it}: ct vs float64 product: max |err| {float(err.max()):.3e}, rows with |err| > 1e-3: {rows_bad.numel()} of {B * S}"
              f" first {rows_bad[:8].tolist()} last {rows_bad[-4:].tolist()}; pre finite {bool(torch.isfinite(pre).all())}", flush=True)
        if rows_bad.numel():
            r = int(rows_bad[0])
            print("   row", r, "got", ct[r, :4].tolist(), "want", want[r, :4].tolist(), "cols bad", (err[r] > 1e-3).nonzero().flatten()[:8].tolist())


if __name__ == "__main__":
    main()
