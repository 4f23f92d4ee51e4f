"""Pin the guide-transformer / residual-VQ oracle (oracle/guide_oracle.py) to the vectors the REFERENCE produced
(tests/golden/make_golden_guide.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from audio2photoreal_amd.spec import GuideSpec, TokenizerSpec
from audio2photoreal_amd.synthetic import synthetic_guide_state_dict, synthetic_tensor, synthetic_tokenizer_state_dict
from conftest import rel_l2
from oracle import guide_oracle as G

SEED = 10
TOL = 2e-5


@pytest.fixture(scope="module")
def gg():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_guide_v1.npz"))


@pytest.fixture(scope="module")
def guide():
    gs = GuideSpec()
    return gs, G.OracleGuide(synthetic_guide_state_dict(gs, SEED), gs.tokens, gs.num_layers, gs.num_heads, gs.audio_conv_dilations)


def _cond(gs, B=2, S=798):
    return synthetic_tensor(SEED, "guide_cond_embed", (B, S, gs.cond_feature_dim))


def test_pre_audio_conv_stack(gg, guide):
    gs, g = guide
    out = g.pre_audio(_cond(gs))
    assert out.shape == (2, gs.cond_tokens_after_conv(798), gs.cond_feature_dim) == (2, 750, 1024)
    assert rel_l2(out[:, ::25], gg["pre_audio_rows25"]) < TOL


def test_teacher_forced_logits_cond_and_uncond(gg, guide):
    gs, g = guide
    toks = torch.from_numpy(gg["fwd/tokens"])
    assert rel_l2(g.forward(toks, _cond(gs)), gg["fwd/logits"]) < 5 * TOL
    assert rel_l2(g.forward(toks, _cond(gs), cond_drop_prob=1.0), gg["fwd/logits_uncond"]) < 5 * TOL


def test_generate_with_injected_uniforms_reproduces_the_reference_tokens(gg, guide):
    gs, g = guide
    u = torch.from_numpy(gg["gen/uniforms"])
    toks = g.generate(_cond(gs), 2, 4, u)
    assert toks.shape == (2, 8) and torch.equal(toks, torch.from_numpy(gg["gen/tokens"]))
    # the nucleus rule on the first step's logits: same support, same renormalised probabilities
    logits = g.forward(torch.full((2, 1), gs.tokens), _cond(gs))[:, -1]
    probs, _ = g.nucleus_probs(logits, 0.94)
    want = torch.from_numpy(gg["gen/sorted_probs"][0])
    assert torch.equal(probs > 0, want > 0) and rel_l2(probs, want) < 1e-4


def test_residual_vq_decode(gg):
    ts = TokenizerSpec()
    out = G.vq_decode(synthetic_tokenizer_state_dict(ts, SEED), torch.from_numpy(gg["vq/tokens"]), ts.residual_depth)
    assert out.shape == (2, 20, ts.n_vertices) and rel_l2(out, gg["vq/decoded"]) < TOL
