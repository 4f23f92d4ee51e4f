"""Which stage of the body path differs between two fresh processes?  (round 4: the pipeline job's body samples were not
reproducible run to run while the face samples were.)  Prints digests of the guide tokens, the VQ-decoded keyframes, the hoisted
audio features and the body samples of one PipelineSubject; run it several times and diff the lines."""
import argparse
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from audio2photoreal_amd.sample import generate as G  # noqa: E402


def h(t):
    return hashlib.sha1(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:12]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--overlap", type=int, default=0)
    ap.add_argument("--subject", type=int, default=1)
    a0 = ap.parse_args()
    a = argparse.Namespace(frames=240, respacing="ddim5", precision="fp16", batch=2)
    dev = torch.device("cuda:0")
    subj = bench.PipelineSubject(a, dev, a0.subject, [0, 1])
    rec = {}
    orig = G._replace_keyframes

    def spy(model_kwargs, model, uniforms=None):
        y = model_kwargs["y"]
        B, T = y["keyframes"].shape[0], y["keyframes"].shape[1]
        with torch.no_grad():
            tokens = model.transformer.generate(y["cond_embed"], T, layers=model.tokenizer.residual_depth, n_sequences=B, max_key_len=T,
                                                max_seq_len=30 * T, uniforms=uniforms)
        rec["tokens"] = h(tokens)
        # guide internals of the hoisted conditioning (valid rows only: S = 798 audio tokens, Sv = S - 2 * sum(dilations))
        import ctypes as C
        import numpy as np
        from audio2photoreal_amd import _lib
        g = model.transformer
        S = y["cond_embed"].shape[1]
        for name, width in (("pre_audio", g.cond_feature_dim), ("ct", 64), ("mem", 64), ("memr", 64), ("hidden", 64), ("film", 6 * 3 * 2 * 64), ("kc", 6 * 64), ("vc", 6 * 64)):
            rows = B if name in ("hidden", "film") else B * S
            host = np.empty((rows, width), np.float32)
            try:
                _lib.check(_lib.load().a2p_guide_debug_read(g._ctx, name.encode(), host.ctypes.data_as(C.c_void_p), host.nbytes), name)
            except Exception as e:   # noqa: BLE001
                rec["g_" + name] = "err"
                continue
            if os.environ.get("A2P_DUMP") and name in ("ct", "pre_audio", "hidden", "mem"):
                np.save(os.path.join(ROOT, "gpurun_out", f"dump_{os.environ['A2P_DUMP']}_{name}.npy"), host)
            if rows == B * S:
                host = host.reshape(B, S, width)[:, : S - 72]        # the rows a valid output can depend on (generous cut)
            rec["g_" + name] = hashlib.sha1(np.ascontiguousarray(host).tobytes()).hexdigest()[:8]
        pred = model.tokenizer.decode(tokens.reshape((B, -1, model.tokenizer.residual_depth))).detach().cpu()
        rec["keyframes"] = h(pred)
        rec["feats"] = h(y["cond_embed"])
        return pred
    G._replace_keyframes = spy
    import audio2photoreal_amd.sample.generate as G2
    G2._replace_keyframes = spy
    for it in range(3):
        if a0.overlap:
            _, body, face = subj.overlapped()
        else:
            _, body, face = subj.sequential()
        print(f"run {it} overlap={a0.overlap}: feats {rec['feats']} tokens {rec['tokens']} keyframes {rec['keyframes']} body {h(body)} face {h(face)} | "
              + " ".join(f"{k[2:]}={v}" for k, v in rec.items() if k.startswith("g_")), flush=True)


if __name__ == "__main__":
    main()
