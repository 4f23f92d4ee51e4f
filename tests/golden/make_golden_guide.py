"""Golden vectors for the guide transformer and the residual-VQ decode, produced by the REFERENCE itself
(model/guide.py, model/vqvae.py from /root/reference, CPU fp32) on the synthetic weights of audio2photoreal_amd.synthetic.
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_guide.py

The audio front end is replaced by given features (as in make_golden.py).  `generate` draws its tokens with
`Categorical(sorted_probs).sample()` from torch's global RNG; here `Categorical` is swapped for an inverse-CDF draw over
injected uniforms that also records the probabilities it was given, so the reference's own loop, nucleus rule included, is what
produces `gen/tokens` and `gen/sorted_probs`.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_import as ri  # noqa: E402
from audio2photoreal_amd.spec import GuideSpec, TokenizerSpec  # noqa: E402
from audio2photoreal_amd.synthetic import (synthetic_guide_state_dict, synthetic_tensor,  # noqa: E402
                                           synthetic_tokenizer_state_dict)

SEED = 10
B, S, STEPS = 2, 798, 8          # 240-frame geometry: 798 audio tokens; 2 keyframes x residual depth 4


def main():
    torch.manual_seed(SEED)
    torch.set_num_threads(8)
    ri.import_reference()
    import model.guide as mg
    import model.vqvae as vq
    gs, ts = GuideSpec(), TokenizerSpec()
    out = {}
    with ri.cpu_cuda(), torch.no_grad():
        g = mg.GuideTransformer(tokens=gs.tokens, num_layers=gs.num_layers, dim=gs.dim, emb_len=gs.emb_len,
                                num_audio_layers=gs.num_audio_layers).eval()
        missing, unexpected = g.load_state_dict(synthetic_guide_state_dict(gs, SEED), strict=False)
        assert not unexpected and all(k.startswith("audio_model.") or k.endswith("rotary.freqs") for k in missing), (missing, unexpected)
        cond = synthetic_tensor(SEED, "guide_cond_embed", (B, S, gs.cond_feature_dim))
        g.encode_audio = lambda raw: cond                                  # features fed past the vq-wav2vec front end
        out["pre_audio_rows25"] = g.pre_audio(cond.permute(0, 2, 1)).permute(0, 2, 1)[:, ::25].contiguous().numpy()   # every 25th of the 750 rows
        toks = torch.from_numpy(np.random.default_rng(SEED).integers(0, gs.tokens, size=(B, 17)))
        toks[:, 0] = gs.tokens
        out["fwd/tokens"] = toks.numpy()
        out["fwd/logits"] = g(toks, cond).numpy()
        out["fwd/logits_uncond"] = g(toks, cond, cond_drop_prob=1.0).numpy()

        uniforms = torch.from_numpy(np.random.default_rng(SEED + 1).random((STEPS, B), dtype=np.float32))
        rec = {"probs": [], "step": 0}

        class InjectedCategorical:
            def __init__(self, probs):
                self.probs = probs

            def sample(self):
                rec["probs"].append(self.probs.clone())
                u = uniforms[rec["step"]]
                rec["step"] += 1
                return (torch.cumsum(self.probs, dim=-1) > u[:, None]).float().argmax(dim=-1)

        mg.Categorical = InjectedCategorical
        tokens = g.generate(cond, STEPS // ts.residual_depth, layers=ts.residual_depth, n_sequences=B, max_key_len=8, max_seq_len=240)
        out["gen/uniforms"], out["gen/tokens"] = uniforms.numpy(), tokens.numpy()
        out["gen/sorted_probs"] = torch.stack(rec["probs"]).numpy()        # [STEPS, B, tokens]

        t = vq.TemporalVertexCodec(n_vertices=ts.n_vertices, latent_dim=ts.latent_dim, categories=ts.categories,
                                   residual_depth=ts.residual_depth).eval()
        missing, unexpected = t.load_state_dict(synthetic_tokenizer_state_dict(ts, SEED), strict=False)
        assert not unexpected, unexpected
        q = torch.from_numpy(np.random.default_rng(SEED + 2).integers(0, ts.categories, size=(B, 20, ts.residual_depth)))
        out["vq/tokens"], out["vq/decoded"] = q.numpy(), t.decode(q).numpy()
    np.savez(os.path.join(HERE, "golden_guide_v1.npz"), **out)
    print({k: (v.dtype, v.shape) for k, v in out.items()}, sum(v.nbytes for v in out.values()) / 1e6, "MB")


if __name__ == "__main__":
    main()
