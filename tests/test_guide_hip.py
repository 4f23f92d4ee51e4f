"""SURVEY.md §8 f2 on the GPU: the guide transformer (hoisted conditioning + one persistent autoregressive launch) and the
residual-VQ decode, through the C ABI, against the vectors the reference produced (tests/golden/make_golden_guide.py).
fp32 throughout: <= 1e-3 relative like the denoiser's parity mode (observed ~1e-6)."""
import os

import numpy as np
import pytest
import torch

from audio2photoreal_amd import _lib
from audio2photoreal_amd.model.guide import GuideTransformer
from audio2photoreal_amd.model.vqvae import TemporalVertexCodec
from audio2photoreal_amd.spec import GuideSpec, TokenizerSpec
from audio2photoreal_amd.synthetic import synthetic_guide_state_dict, synthetic_tensor, synthetic_tokenizer_state_dict
from conftest import rel_l2, rel_max

pytestmark = pytest.mark.gpu
SEED = 10


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def gg():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_guide_v1.npz"))


@pytest.fixture(scope="module")
def guide(dev):
    gs = GuideSpec()
    g = GuideTransformer(tokens=gs.tokens, num_layers=gs.num_layers, dim=gs.dim, emb_len=gs.emb_len,
                         num_audio_layers=gs.num_audio_layers, max_batch=4, max_positions=96)
    g.load_state_dict(synthetic_guide_state_dict(gs, SEED), strict=False)
    return gs, g.to(dev).eval()


def _cond(gs, dev, B=2, S=798):
    return synthetic_tensor(SEED, "guide_cond_embed", (B, S, gs.cond_feature_dim)).to(dev)


def test_teacher_forced_logits_and_hoisted_conv_stack_vs_reference(dev, gg, guide):
    gs, g = guide
    cond = _cond(gs, dev)
    toks = torch.from_numpy(gg["fwd/tokens"]).to(dev)
    logits = g(toks, cond)
    e = rel_l2(logits.cpu(), gg["fwd/logits"])
    print(f"guide logits rel L2 {e:.3e}, max-norm {rel_max(logits.cpu(), gg['fwd/logits']):.3e}")
    assert e < 1e-3 and rel_max(logits.cpu(), gg["fwd/logits"]) < 1e-3
    rows = g.pre_audio_features(2 * 798).view(2, 798, -1)[:, :750]          # valid rows of each sequence (row stride S)
    assert rel_l2(rows[:, ::25], gg["pre_audio_rows25"]) < 1e-3
    unc = g(toks, cond, cond_drop_prob=1.0)
    assert rel_l2(unc.cpu(), gg["fwd/logits_uncond"]) < 1e-3
    again = g(toks, cond)                                                    # re-prepare after the unconditional pass
    assert torch.equal(again, logits)


def test_generate_reproduces_the_reference_tokens_and_nucleus_probabilities(dev, gg, guide):
    gs, g = guide
    u = torch.from_numpy(gg["gen/uniforms"]).to(dev)
    toks, probs = g.generate(_cond(gs, dev), 2, 4, n_sequences=2, uniforms=u, return_probs=True)
    assert toks.dtype == torch.int64 and toks.shape == (2, 8)
    assert torch.equal(toks.cpu(), torch.from_numpy(gg["gen/tokens"]))
    want = torch.from_numpy(gg["gen/sorted_probs"])                           # [steps, B, tokens]
    assert torch.equal(probs.cpu() > 0, want > 0)                             # same nucleus at every step
    assert rel_l2(probs.cpu(), want) < 1e-3


def test_full_length_generate_and_properties(dev, guide):
    """600-frame geometry: 1998 audio tokens, 20 keyframes x depth 4 = 80 steps, 4 sequences in one launch."""
    gs, g = guide
    cond = synthetic_tensor(SEED, "guide_cond_full", (4, 1998, gs.cond_feature_dim)).to(dev)
    u = torch.rand(80, 4, generator=torch.Generator().manual_seed(3)).to(dev)
    a = g.generate(cond, 20, 4, n_sequences=4, max_key_len=20, max_seq_len=600, uniforms=u)
    b = g.generate(cond, 20, 4, n_sequences=4, max_key_len=20, max_seq_len=600, uniforms=u)
    assert a.shape == (4, 80) and torch.equal(a, b) and int(a.min()) >= 0 and int(a.max()) < gs.tokens
    # teacher forcing on the sampled prefix reproduces the distribution the sampler saw: its argmax is the token drawn at u -> 0
    greedy = g.generate(cond, 20, 4, n_sequences=4, max_key_len=20, max_seq_len=600, uniforms=torch.zeros_like(u))
    prefix = torch.cat([torch.full((4, 1), gs.tokens, device=dev), greedy[:, :-1]], dim=1)
    assert torch.equal(g(prefix, cond).argmax(-1), greedy)
    # sequences are independent: sequence 2 alone gives the same tokens
    solo = g.generate(cond[2:3].contiguous(), 20, 4, n_sequences=1, max_key_len=20, max_seq_len=600, uniforms=u[:, 2:3].contiguous())
    assert torch.equal(solo[0], a[2])


def test_residual_vq_decode_vs_reference(dev, gg):
    ts = TokenizerSpec()
    t = TemporalVertexCodec(ts.n_vertices, ts.latent_dim, ts.categories, ts.residual_depth)
    t.load_state_dict(synthetic_tokenizer_state_dict(ts, SEED), strict=False)
    out = t.to(dev).decode(torch.from_numpy(gg["vq/tokens"]).to(dev))
    assert out.shape == (2, 20, ts.n_vertices) and rel_l2(out.cpu(), gg["vq/decoded"]) < 1e-4


def test_guide_error_paths(dev, guide):
    gs, g = guide
    with pytest.raises(_lib.A2PError):
        g.generate(_cond(gs, dev, B=2), 40, 4, n_sequences=2, max_key_len=40, max_seq_len=1200)      # 160 > max_positions
    with pytest.raises(_lib.A2PError):
        g(torch.zeros(2, 4, dtype=torch.int64, device=dev), torch.zeros(2, 40, gs.cond_feature_dim, device=dev))  # no rows left after the convs
    with pytest.raises(_lib.A2PError):
        g.encode_audio(torch.zeros(1, 16000, 2, device=dev))


def test_replace_keyframes_feeds_the_pose_denoiser(dev, guide):
    """sample/generate.py:51-71 surface: guide tokens -> VQ decode -> `y["keyframes"]` of the body model, end to end."""
    from types import SimpleNamespace
    from audio2photoreal_amd.model.cfg_sampler import ClassifierFreeSampleModel
    from audio2photoreal_amd.model_util import create_model_and_diffusion, default_args, load_model
    from audio2photoreal_amd.sample.generate import _replace_keyframes, _run_single_diffusion
    from audio2photoreal_amd.spec import pose_spec
    from audio2photoreal_amd.synthetic import synthetic_inputs, synthetic_state_dict
    gs, g = guide
    ts = TokenizerSpec()
    tok = TemporalVertexCodec(ts.n_vertices, ts.latent_dim, ts.categories, ts.residual_depth)
    tok.load_state_dict(synthetic_tokenizer_state_dict(ts, SEED), strict=False)
    spec = pose_spec()
    model, diffusion = create_model_and_diffusion(default_args("pose", timestep_respacing="ddim10"), "test", precision="fp32", max_batch=2)
    load_model(model, synthetic_state_dict(spec, SEED))
    model.setup_guide_predictor(g, tok.to(dev))
    assert not any(k.startswith(("transformer.", "tokenizer.")) for k in model._hot_state())
    load_model(model, synthetic_state_dict(spec, SEED))          # `transformer.` / `tokenizer.` keys may be missing (reference rule)
    cfg = ClassifierFreeSampleModel(model.to(dev).eval())
    assert cfg.transformer is g and cfg.tokenizer is tok
    inp = synthetic_inputs(spec, 2, 240, SEED)
    y = {"cond_embed": inp["cond_embed"].to(dev), "keyframes": inp["keyframes"].to(dev), "mask": inp["mask"].to(dev),
         "scale": torch.full((2,), 2.0, device=dev)}
    u = torch.rand(8 * ts.residual_depth, 2, generator=torch.Generator().manual_seed(5)).to(dev)
    pred = _replace_keyframes({"y": y}, cfg, uniforms=u)
    assert pred.shape == y["keyframes"].shape == (2, 8, 104) and bool(torch.isfinite(pred).all())
    toks = g.generate(y["cond_embed"], 8, ts.residual_depth, n_sequences=2, uniforms=u)
    assert torch.equal(pred, tok.decode(toks.reshape(2, -1, ts.residual_depth)).cpu())
    args = SimpleNamespace(data_format="pose", resume_trans="ckpt", batch_size=2, curr_seq_length=240)
    sample, _, kf, _ = _run_single_diffusion(args, {"y": y}, diffusion, cfg, lambda v, kind: v, None, noise=inp["x_T"].to(dev))
    assert sample.shape == (2, 104, 1, 240) and bool(torch.isfinite(sample).all()) and kf.shape == (2, 8, 104)
