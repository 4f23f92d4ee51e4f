"""CPU: the front-end oracle (oracle/frontend_oracle.py) against the fixtures the REFERENCE produced
(tests/golden/make_golden_frontend.py -> golden_frontend_v1.npz), and the host-side pieces of the native front end."""
import os

import numpy as np
import pytest
import torch

from audio2photoreal_amd.model.audio_frontend import CONV_GEOMETRY, NativeAudioFrontend
from audio2photoreal_amd.synthetic import cond_tokens_for_frames, synthetic_audio, synthetic_frontend_state_dict
from conftest import ROOT, rel_l2
from oracle import frontend_oracle as FO

SEED, B, FRAMES = 10, 2, 240


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_frontend_v1.npz"))


def test_oracle_front_end_matches_the_reference(gold):
    sd = synthetic_frontend_state_dict(SEED, lip=True)
    audio = synthetic_audio(SEED, B, FRAMES)
    with torch.no_grad():
        emb = FO.encode_audio(audio, sd)
        lip = FO.lip_frames(audio, sd)
        full = FO.encode_lip(audio, emb, sd)
    assert tuple(full.shape) == tuple(gold["shape"]) == (B, 798, 2038)
    assert rel_l2(emb[:, ::16], gold["emb_rows16"]) < 1e-5 and abs(float(emb.norm()) / float(gold["emb_norm"]) - 1) < 1e-5
    assert rel_l2(lip[:, ::4].reshape(B, -1, 1014), gold["lip_frames4"]) < 1e-4
    assert rel_l2(full[:, ::16], gold["full_rows16"]) < 1e-4 and abs(float(full.norm()) / float(gold["full_norm"]) - 1) < 1e-4


def test_token_geometry_matches_the_reference_constants():
    # model/diffusion.py:136 hard-codes 1998 tokens for 600 frames; train/train_guide.py:316 has 798 for 240
    assert NativeAudioFrontend.n_tokens(600 * 1600) == 1998 == cond_tokens_for_frames(600)
    assert NativeAudioFrontend.n_tokens(240 * 1600) == 798 == cond_tokens_for_frames(240)
    assert len(CONV_GEOMETRY) == 8


def test_sinc_resampler_restatement_is_a_unit_gain_lowpass():
    """No torchaudio here to pin against ("parity unpinned"): properties of the documented kernel instead -- DC gain 1, a 1 kHz
    tone passes, a 12 kHz tone (above the 8 kHz Nyquist of the output) is removed, output length ceil(L / 3)."""
    n = torch.arange(4800, dtype=torch.float32)
    dc = FO.resample_sinc(torch.ones(1, 4801))
    assert dc.shape[-1] == 1601 and abs(float(dc[0, 100:1500].mean()) - 1.0) < 1e-3
    lo = FO.resample_sinc(torch.sin(2 * torch.pi * 1000 / 48000 * n)[None])
    hi = FO.resample_sinc(torch.sin(2 * torch.pi * 12000 / 48000 * n)[None])
    want = torch.sin(2 * torch.pi * 1000 / 16000 * torch.arange(1600, dtype=torch.float32))
    assert float((lo[0, 50:1550] - want[50:1550]).abs().max()) < 5e-3
    assert float(hi[0, 50:1550].abs().max()) < 5e-3


def test_state_dict_layout_of_the_native_front_end():
    from audio2photoreal_amd.model_util import create_model_and_diffusion, default_args, load_model
    from audio2photoreal_amd.spec import face_spec
    from audio2photoreal_amd.synthetic import synthetic_state_dict
    model, _ = create_model_and_diffusion(default_args("face", layers=1), "test", audio_frontend="native")
    keys = set(model.state_dict())
    fe = synthetic_frontend_state_dict(SEED, lip=True)
    assert set(fe) <= keys                                     # audio_model.* / lip_model.* under the reference's names
    load_model(model, {**synthetic_state_dict(face_spec(num_layers=1), SEED), **fe,
                       "audio_model.feature_aggregator.conv_layers.0.0.weight": torch.zeros(3)})   # off-path tensors are skipped
    assert torch.equal(model.lip_model.project_output.weight, fe["lip_model.project_output.weight"])
    pose, _ = create_model_and_diffusion(default_args("pose", layers=1), "test", audio_frontend="native")
    assert not hasattr(pose, "lip_model") and hasattr(pose, "audio_model")


def test_fairseq_block_restatement_equals_a_module_composition():
    """oracle/frontend_oracle.py restates fairseq's ConvFeatureExtractionModel / ConvAggregator functionally (PARITY UNPINNED: the
    package is absent).  The same published block definitions composed from torch.nn MODULES (Conv1d, GroupNorm(1, C),
    ReplicationPad1d / ConstantPad1d, GELU / ReLU) must give the same features: guards the functional form -- padding side and
    width, residual subsampling, where the scale and the log sit -- against slips, not against fairseq itself."""
    import dataclasses
    import math
    import torch.nn as nn
    from audio2photoreal_amd.model.audio_frontend import FAIRSEQ
    geo = dataclasses.replace(FAIRSEQ, l_skip=True, agg_layers=4)
    sd = synthetic_frontend_state_dict(SEED, lip=True, geometry=geo)
    x = torch.randn(2, 4000, generator=torch.Generator().manual_seed(3))
    P = "lip_model.audio_encoder.wav2vec_model."
    with torch.no_grad():
        got = FO.conv_features(x, sd, P + "feature_extractor.", layers=7, group_norm=True, activation="relu", log_compression=True, skip=True,
                               residual_scale=0.5)
        h = x.unsqueeze(1)
        for i, (k, s) in enumerate(CONV_GEOMETRY[:7]):
            conv = nn.Conv1d(h.shape[1], 512, k, stride=s, bias=False)
            gn = nn.GroupNorm(1, 512)
            conv.weight.copy_(sd[f"{P}feature_extractor.conv_layers.{i}.0.weight"])
            gn.weight.copy_(sd[f"{P}feature_extractor.conv_layers.{i}.2.weight"]); gn.bias.copy_(sd[f"{P}feature_extractor.conv_layers.{i}.2.bias"])
            res = h
            h = nn.ReLU()(gn(conv(h)))
            if h.shape[1] == res.shape[1]:
                h = (h + res[..., :: res.shape[2] // h.shape[2]][..., : h.shape[2]]) * math.sqrt(0.5)
        want = (h.abs() + 1).log()
        assert got.shape == want.shape and rel_l2(got, want) < 1e-6
        agg = FO.conv_aggregator(got, sd, P + "feature_aggregator.", 4, True, 0.5, True, False, "relu")
        h = want
        for j in range(4):
            k = j + 2
            conv = nn.Conv1d(512, 512, k)
            gn = nn.GroupNorm(1, 512)
            conv.weight.copy_(sd[f"{P}feature_aggregator.conv_layers.{j}.1.weight"]); conv.bias.copy_(sd[f"{P}feature_aggregator.conv_layers.{j}.1.bias"])
            gn.weight.copy_(sd[f"{P}feature_aggregator.conv_layers.{j}.3.weight"]); gn.bias.copy_(sd[f"{P}feature_aggregator.conv_layers.{j}.3.bias"])
            h = (nn.ReLU()(gn(conv(nn.ReplicationPad1d((k - 1, 0))(h)))) + h) * math.sqrt(0.5)
        assert agg.shape == h.shape == got.shape and rel_l2(agg, h) < 1e-6
