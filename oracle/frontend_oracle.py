"""ORACLE -- test infrastructure, NOT product code (same rules as oracle/a2p_oracle.py).

CPU restatement (torch CPU fp32) of the reference's audio front end: `FiLMTransformer.encode_audio` /
`encode_lip` (model/diffusion.py:285-313), `Audio2LipRegressionTransformer` (:37-79), `Wav2VecEncoder`
(model/modules/audio_encoder.py:24-46), `RegressionTransformer` and its blocks (model/modules/transformer_modules.py:281-303,
351-512, 560-627).  Paths relative to /root/reference.

Pinning: `tests/golden/make_golden_frontend.py` ran the reference's own `encode_audio` / `encode_lip` (fairseq / torchaudio
stubbed per SURVEY.md Appendix A: bias-free conv + ReLU stack with the vq-wav2vec geometry, x[::3] resampler) on
audio2photoreal_amd.synthetic weights; `tests/test_frontend_oracle.py` checks this file against those fixtures.
PARITY UNPINNED for the two third-party pieces: fairseq's real (vq-)wav2vec models and torchaudio's Resample, restated below
from their published sources with nothing in the container to check them against:
  * `resample_sinc`: torchaudio 2.0.2 functional `_get_sinc_resample_kernel` / `_apply_sinc_resample_kernel`;
  * `conv_features(..., geometry)` / `conv_aggregator` (round 4): fairseq 0.12 fairseq/models/wav2vec/wav2vec.py --
    `ConvFeatureExtractionModel` (block = Conv1d(bias=False) -> Dropout -> Fp32GroupNorm(1, C) -> activation; forward: skip
    connections `(x + residual[..., ::r][..., :T]) * sqrt(residual_scale)`, then `log(|x| + 1)` with --log-compression) and
    `ConvAggregator` (block = ReplicationPad1d((k - 1, 0)) | ZeroPad1d -> Conv1d(k) -> Dropout -> Fp32GroupNorm(1, C) -> activation;
    forward: `x = (block(x) + residual) * sqrt(residual_scale)` with --skip-connections-agg).  `geometry` is any object with the
    attributes of audio2photoreal_amd.model.audio_frontend.FrontendGeometry (a_* / l_* / agg_*); None = the stub geometry.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

from .a2p_oracle import layer_norm, mha

Tensor = torch.Tensor
CONV_GEOMETRY = ((10, 5), (8, 4), (4, 2), (4, 2), (4, 2), (1, 1), (1, 1), (1, 1))


def resample_decimate(x: Tensor) -> Tensor:
    """The golden generator's stand-in for torchaudio Resample(48000, 16000): x[..., ::3] (tests/golden/ref_import.py)."""
    return x[..., ::3]


def resample_sinc(x: Tensor) -> Tensor:
    """torchaudio.transforms.Resample(48000, 16000) defaults (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99), restated from
    torchaudio.functional's `_get_sinc_resample_kernel` / `_apply_sinc_resample_kernel` (torchaudio==2.0.2 is pinned by
    scripts/requirements.txt:15 and absent offline): gcd-reduced rates 3 -> 1, width = ceil(6 * 3 / 0.99) = 19, 41 taps,
    conv1d(pad(x, (19, 22)), kernel, stride 3)[..., : ceil(L / 3)]."""
    orig, new, lpw, rolloff = 3, 1, 6, 0.99
    base = min(orig, new) * rolloff
    width = math.ceil(lpw * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = (torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx) * base
    t = t.clamp(-lpw, lpw)
    window = torch.cos(t * math.pi / lpw / 2) ** 2
    t = t * math.pi
    kernel = (torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / orig)).to(torch.float32)
    shape = x.shape
    w = F.pad(x.reshape(-1, shape[-1]), (width, width + orig))
    y = F.conv1d(w[:, None], kernel, stride=orig).transpose(1, 2).reshape(w.shape[0], -1)
    return y[..., : math.ceil(new * shape[-1] / orig)].reshape(*shape[:-1], -1)


def _act(name: str):
    return F.gelu if name == "gelu" else F.relu


def conv_features(x: Tensor, sd: Dict[str, Tensor], prefix: str, layers: int = 8, group_norm: bool = False, activation: str = "relu",
                  log_compression: bool = False, skip: bool = False, residual_scale: float = 0.5) -> Tensor:
    """`model.feature_extractor(x)`: x [B, L] -> [B, 512, L'].  Defaults = the stub geometry (8 x (Conv1d(bias=False), ReLU));
    the options are fairseq's ConvFeatureExtractionModel (module docstring)."""
    h = x.unsqueeze(1)
    act, rs = _act(activation), math.sqrt(residual_scale)
    for i, (_, s) in enumerate(CONV_GEOMETRY[:layers]):
        residual = h
        h = F.conv1d(h, sd[f"{prefix}conv_layers.{i}.0.weight"], stride=s)
        if group_norm:
            h = F.group_norm(h.float(), 1, sd[f"{prefix}conv_layers.{i}.2.weight"], sd[f"{prefix}conv_layers.{i}.2.bias"], 1e-5)
        h = act(h)
        if skip and h.size(1) == residual.size(1):
            tsz, r_tsz = h.size(2), residual.size(2)
            residual = residual[..., :: r_tsz // tsz][..., :tsz]
            h = (h + residual) * rs
    if log_compression:
        h = (h.abs() + 1).log()
    return h


def conv_aggregator(x: Tensor, sd: Dict[str, Tensor], prefix: str, layers: int, skip: bool = True, residual_scale: float = 0.5,
                    conv_bias: bool = True, zero_pad: bool = False, activation: str = "relu") -> Tensor:
    """fairseq ConvAggregator.forward on x [B, 512, T] (all layers 512 -> 512: residual_proj is None everywhere)."""
    act, rs = _act(activation), math.sqrt(residual_scale)
    for j in range(layers):
        k = j + 2
        ka = k // 2
        kb = ka - 1 if k % 2 == 0 else ka
        residual = x
        h = F.pad(x, (ka + kb, 0), mode="constant" if zero_pad else "replicate")
        h = F.conv1d(h, sd[f"{prefix}conv_layers.{j}.1.weight"], sd[f"{prefix}conv_layers.{j}.1.bias"] if conv_bias else None)
        h = F.group_norm(h.float(), 1, sd[f"{prefix}conv_layers.{j}.3.weight"], sd[f"{prefix}conv_layers.{j}.3.bias"], 1e-5)
        x = act(h)
        if skip:
            x = (x + residual) * rs
    return x


def _stack_kw(geometry, side: str) -> dict:
    if geometry is None:
        return {}
    g = lambda n: getattr(geometry, f"{side}_{n}")
    return dict(layers=(geometry.l_layers if side == "l" else 8), group_norm=g("group_norm"), activation=g("activation"),
                log_compression=g("log_compression"), skip=g("skip"), residual_scale=g("residual_scale"))


def encode_audio(audio: Tensor, sd: Dict[str, Tensor], resample=resample_decimate, geometry=None) -> Tensor:
    """model/diffusion.py:285-293: both channels resampled, feature-extracted, concatenated -> [B, S, 1024]."""
    kw = _stack_kw(geometry, "a")
    z0 = conv_features(resample(audio[:, :, 0]), sd, "audio_model.feature_extractor.", **kw)
    z1 = conv_features(resample(audio[:, :, 1]), sd, "audio_model.feature_extractor.", **kw)
    return torch.cat((z0, z1), dim=1).permute(0, 2, 1)


def wav2vec_encoder(audio: Tensor, sd: Dict[str, Tensor], resample, geometry=None) -> Tensor:
    """Wav2VecEncoder.forward (audio_encoder.py:34-46): [B, T, 1600] -> resample -> 320 zeros on the left -> feature extractor
    -> feature aggregator (the identity in the stub geometry) -> [B, T_w, 512]."""
    a = resample(audio.reshape(audio.shape[0], -1))
    a = torch.cat([torch.zeros(a.shape[0], 320), a], dim=-1)
    W = "lip_model.audio_encoder.wav2vec_model."
    x = conv_features(a, sd, W + "feature_extractor.", **_stack_kw(geometry, "l"))
    if geometry is not None and geometry.agg_layers:
        x = conv_aggregator(x, sd, W + "feature_aggregator.", geometry.agg_layers, geometry.agg_skip, geometry.agg_residual_scale,
                            geometry.agg_conv_bias, geometry.agg_zero_pad, geometry.agg_activation)
    return x.permute(0, 2, 1).contiguous()


def regression_transformer(x: Tensor, cond: Tensor, sd: Dict[str, Tensor], heads: int = 4) -> Tensor:
    """RegressionTransformer.forward, causal=False (transformer_modules.py:594-627): positional encodings, pre-norm encoder layers
    over the condition (:449-472), pre-norm decoder layers with self / cross attention and a ReLU feed-forward (:475-512)."""
    R = "lip_model.regression_model."
    g = lambda n: sd[R + n]
    x = x + g("target_positional_encoding.pe")[None, : x.shape[1]]
    cond = cond + g("cond_positional_encoding.pe")[None, : cond.shape[1]]

    def attn(q, kv, p):
        return mha(q, kv, kv, g(p + ".in_proj_weight"), g(p + ".in_proj_bias"), g(p + ".out_proj.weight"), g(p + ".out_proj.bias"), heads)

    def ffn(h, p):
        return F.relu(h @ g(p + ".ff.0.weight").T + g(p + ".ff.0.bias")) @ g(p + ".ff.3.weight").T + g(p + ".ff.3.bias")

    i = 0
    while R + f"transformer_encoder.{i}.norm1.weight" in sd:
        p = f"transformer_encoder.{i}."
        h = layer_norm(cond, g(p + "norm1.weight"), g(p + "norm1.bias"))
        cond = cond + attn(h, h, p + "self_attn.self_attn")
        cond = cond + ffn(layer_norm(cond, g(p + "norm2.weight"), g(p + "norm2.bias")), p + "feedforward")
        i += 1
    i = 0
    while R + f"transformer_decoder.{i}.norm1.weight" in sd:
        p = f"transformer_decoder.{i}."
        h = layer_norm(x, g(p + "norm1.weight"), g(p + "norm1.bias"))
        x = x + attn(h, h, p + "self_attn.self_attn")
        x = x + attn(layer_norm(x, g(p + "norm2.weight"), g(p + "norm2.bias")), cond, p + "cross_attn.cross_attn")
        x = x + ffn(layer_norm(x, g(p + "norm3.weight"), g(p + "norm3.bias")), p + "feedforward")
        i += 1
    return x


def lip_model(audio: Tensor, sd: Dict[str, Tensor], resample, geometry=None) -> Tensor:
    """Audio2LipRegressionTransformer.forward (model/diffusion.py:63-79): [B, T, 1600] -> [B, T, 338, 3]."""
    B, T = audio.shape[0], audio.shape[1]
    cond = wav2vec_encoder(audio, sd, resample, geometry)
    x = regression_transformer(torch.zeros(B, T, 512), cond, sd)
    x = x @ sd["lip_model.project_output.weight"].T + sd["lip_model.project_output.bias"]
    return x.view(B, T, -1, 3)


def lip_frames(audio: Tensor, sd: Dict[str, Tensor], resample=resample_decimate, geometry=None) -> Tensor:
    """First half of encode_lip (model/diffusion.py:296-306): channel 0 in 120-frame chunks through the lip model -> [B, T, 338, 3]."""
    reshaped = audio.reshape((audio.shape[0], -1, 1600, 2))[..., 0]
    B, T, _ = reshaped.shape
    lip = torch.zeros((B, T, 338, 3))
    for i in range(0, T, 120):
        lip[:, i: i + 120] = lip_model(reshaped[:, i: i + 120], sd, resample, geometry)
    return lip


def encode_lip(audio: Tensor, cond_embed: Tensor, sd: Dict[str, Tensor], resample=resample_decimate, geometry=None) -> Tensor:
    """model/diffusion.py:295-313."""
    lip = lip_frames(audio, sd, resample, geometry)
    B, T = lip.shape[0], lip.shape[1]
    lip = lip.permute(0, 2, 3, 1).reshape((B, 338 * 3, -1))
    lip = F.interpolate(lip, size=cond_embed.shape[1], mode="nearest-exact").permute(0, 2, 1)
    return torch.cat((cond_embed, lip), dim=-1)
