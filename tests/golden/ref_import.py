"""Import the read-only reference (/root/reference) in THIS container only.

Test/fixture infrastructure (never imported by the product).  Follows the
recipe of SURVEY.md Appendix A: stub `fairseq` / `torchaudio` (absent, no
network), skip the absent lip-regressor checkpoint (model/diffusion.py:273-280),
neutralise the hard-coded `.cuda()` (model/diffusion.py:321) and restore the
undefined `noise` in GaussianDiffusion.p_sample (gaussian_diffusion.py:476).
"""
import argparse
import os
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference"


def have_reference() -> bool:
    return os.path.isdir(os.path.join(REF, "model"))


class _StubWav2Vec(nn.Module):
    """vq-wav2vec conv geometry (k,s)=(10,5),(8,4),(4,2)x3,(1,1)x3, 512 ch."""

    def __init__(self):
        super().__init__()
        ks = [(10, 5), (8, 4), (4, 2), (4, 2), (4, 2), (1, 1), (1, 1), (1, 1)]
        layers, cin = [], 1
        for k, s in ks:
            layers += [nn.Conv1d(cin, 512, k, stride=s, bias=False), nn.ReLU()]
            cin = 512
        self.net = nn.Sequential(*layers)

    def feature_extractor(self, x):
        return self.net(x.unsqueeze(1))

    def feature_aggregator(self, x):
        return x


class _StubResample(nn.Module):
    def __init__(self, orig_freq, new_freq):
        super().__init__()
        self.r = orig_freq // new_freq

    def forward(self, x):
        return x[..., :: self.r]


def install_stubs():
    if "fairseq" not in sys.modules:
        fs = types.ModuleType("fairseq")
        cu = types.ModuleType("fairseq.checkpoint_utils")
        cu.load_model_ensemble_and_task = lambda paths: ([_StubWav2Vec()], None, None)
        fs.checkpoint_utils = cu
        sys.modules["fairseq"] = fs
        sys.modules["fairseq.checkpoint_utils"] = cu
    if "torchaudio" not in sys.modules:
        ta = types.ModuleType("torchaudio")
        tr = types.ModuleType("torchaudio.transforms")
        tr.Resample = _StubResample
        ta.transforms = tr
        sys.modules["torchaudio"] = ta
        sys.modules["torchaudio.transforms"] = tr


def import_reference():
    """Returns a namespace with the reference modules (patched for CPU)."""
    assert have_reference(), "reference tree not present (only in the build container)"
    sys.dont_write_bytecode = True
    install_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import model.diffusion as md
    import model.cfg_sampler as cfg
    import diffusion.gaussian_diffusion as gd
    import diffusion.respace as rs
    import utils.model_util as mu

    def _setup_lip_models(self):
        # checkpoint ./assets/iter-0200000.pt is absent; the lip model is a
        # conditioning producer outside the hot path
        self.lip_model = nn.Identity()

    # the reference's own front end, kept reachable for tests/golden/make_golden_frontend.py
    if not hasattr(md.FiLMTransformer, "_ref_encode_audio"):
        md.FiLMTransformer._ref_encode_audio = md.FiLMTransformer.encode_audio
        md.FiLMTransformer._ref_encode_lip = md.FiLMTransformer.encode_lip
    md.FiLMTransformer.setup_lip_models = _setup_lip_models

    # decoder-only conditioning: feed precomputed features
    def _encode_audio(self, raw_audio):
        return self._a2p_cond_embed

    def _encode_lip(self, audio, cond_embed):
        return cond_embed

    md.FiLMTransformer.encode_audio = _encode_audio
    md.FiLMTransformer.encode_lip = _encode_lip

    # restore the undefined `noise` of p_sample (gaussian_diffusion.py:476):
    # noise = randn_like(x), taken from an injected list when present.
    def _p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None,
                  cond_fn=None, model_kwargs=None, const_noise=False):
        out = self.p_mean_variance(model, x, t, clip_denoised=clip_denoised,
                                   denoised_fn=denoised_fn, model_kwargs=model_kwargs)
        inj = getattr(self, "_a2p_step_noise", None)
        noise = inj.pop(0) if inj else torch.randn_like(x)
        if const_noise:
            noise = noise[[0]].repeat(x.shape[0], 1, 1, 1)
        nonzero_mask = (t != 0).float().view(-1, *([1] * (len(x.shape) - 1)))
        sample = out["mean"] + nonzero_mask * torch.exp(0.5 * out["log_variance"]) * noise
        return {"sample": sample, "pred_xstart": out["pred_xstart"]}

    gd.GaussianDiffusion.p_sample = _p_sample
    return types.SimpleNamespace(md=md, cfg=cfg, gd=gd, rs=rs, mu=mu)


def ref_args(data_format, layers, heads, timestep_respacing="ddim10", max_seq_length=600):
    return argparse.Namespace(
        data_format=data_format, layers=layers, heads=heads,
        add_frame_cond=1 if data_format == "pose" else None,
        max_seq_length=max_seq_length, not_rotary=False, unconstrained=False,
        device="cpu", timestep_respacing=timestep_respacing, noise_schedule="cosine",
        sigma_small=True, lambda_vel=0.0, model_path="x", resume_trans=None)


class _cpu_cuda:
    """`.cuda()` -> identity while building/running the reference on CPU."""

    def __enter__(self):
        self._orig = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        return self

    def __exit__(self, *exc):
        torch.Tensor.cuda = self._orig


def build_reference_model(ns, data_format, layers, heads, timestep_respacing):
    args = ref_args(data_format, layers, heads, timestep_respacing)
    model, diffusion = ns.mu.create_model_and_diffusion(args, split_type="test")
    model.eval()
    return model, diffusion


cpu_cuda = _cpu_cuda
