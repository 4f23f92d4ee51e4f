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
        print(f"run {it} overlap={a0.overlap}: feats {rec['feats']} tokens {rec['tokens']} keyframes {rec['keyframes']} body {h(body)} face {h(face)}", flush=True)


if __name__ == "__main__":
    main()
