"""Deterministic synthetic weights / inputs (no checkpoints or datasets exist
offline: SURVEY.md §8c "Pretrained weights / real data: Absent").

Values come from numpy's PCG64 stream keyed by (seed, parameter name) so the
same tensors can be rebuilt bit-for-bit on the GPU box, in the golden-vector
generator (tests/golden/make_golden.py, which loads them into the *reference*
modules) and in bench.py.  Scales follow the reference initialisers in spirit
(xavier-normal for Linear/Conv weights, model/utils.py:29-38; randn null
embeddings, model/diffusion.py:137-138) with non-trivial biases / LayerNorm
affine terms so every term of every formula is exercised.
"""
from __future__ import annotations

import zlib
from typing import Dict

import numpy as np
import torch

from .spec import DenoiserSpec, GuideSpec, TokenizerSpec, guide_param_shapes, param_shapes, tokenizer_param_shapes


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def synthetic_tensor(seed: int, name: str, shape, scale: float = 1.0, shift: float = 0.0) -> torch.Tensor:
    a = _rng(seed, name).standard_normal(size=tuple(shape), dtype=np.float64)
    return torch.from_numpy((a * scale + shift).astype(np.float32))


def synthetic_state_dict(spec: DenoiserSpec, seed: int = 10) -> Dict[str, torch.Tensor]:
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(spec).items():
        leaf = name.split(".")[-1]
        is_norm = (".norm" in name or name.startswith("norm_cond") or name.startswith("frame_norm_cond")
                   or name.startswith("non_attn_cond_projection.0"))
        if name.startswith("null_"):
            t = synthetic_tensor(seed, name, shape, 1.0)
        elif is_norm and leaf == "weight":
            t = synthetic_tensor(seed, name, shape, 0.1, 1.0)
        elif is_norm and leaf == "bias":
            t = synthetic_tensor(seed, name, shape, 0.1)
        elif leaf in ("bias", "in_proj_bias"):
            t = synthetic_tensor(seed, name, shape, 0.02)
        else:  # Linear / Conv / in_proj weights: xavier-normal std
            if len(shape) == 3:
                fan_out, fan_in = shape[0] * shape[2], shape[1] * shape[2]
            else:
                fan_out, fan_in = shape[0], shape[1]
            t = synthetic_tensor(seed, name, shape, float(np.sqrt(2.0 / (fan_in + fan_out))))
        sd[name] = t
    # rotary frequency buffer (model/modules/rotary_embedding_torch.py:99-101)
    d = spec.latent_dim
    sd["rotary.freqs"] = 1.0 / (10000 ** (torch.arange(0, d, 2)[: d // 2].float() / d))
    return sd


def trained_like_state_dict(spec: DenoiserSpec, seed: int = 10, weight_gain: float = 1.0, qk_gain: float = 1.0,
                            resid_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """synthetic_state_dict pushed towards the statistics of a TRAINED denoiser (xavier-scale weights give near-uniform softmax
    rows and a residual stream of a few units: the easy case for 16-bit operands).  `weight_gain` multiplies every Linear / conv
    weight of the decoder stack, the conditioning path and the output head (activations grow with depth); `qk_gain` multiplies
    the query and key rows of every attention in_proj on top of that (logits grow with its square: peaky softmax rows);
    `resid_gain` multiplies input_projection (weight and bias): a residual stream of that many units enters layer 0, rides through
    every LayerNorm / FiLM / out_proj epilogue and is what final_layer's split-operand rows have to carry."""
    sd = synthetic_state_dict(spec, seed)
    d = spec.latent_dim
    for name, t in sd.items():
        leaf = name.split(".")[-1]
        if name == "rotary.freqs" or name.startswith("null_") or ".norm" in name or name.startswith("norm_cond") or name.startswith("frame_norm_cond"):
            continue
        if leaf in ("weight", "in_proj_weight") and t.dim() >= 2:
            t = t * weight_gain
            if leaf == "in_proj_weight" and qk_gain != 1.0:
                t = t.clone()
                t[: 2 * d] *= qk_gain
            sd[name] = t
    if resid_gain != 1.0:
        for k in ("input_projection.weight", "input_projection.bias"):
            sd[k] = sd[k] * resid_gain
    return sd


def _init_like_reference(seed: int, name: str, shape, prefix: str = "") -> torch.Tensor:
    """The rule synthetic_state_dict applies per parameter, shared with the guide / tokenizer dictionaries (`prefix` only
    separates their random streams)."""
    leaf, key = name.split(".")[-1], prefix + name
    is_norm = ".norm" in name or name.startswith("norm_cond") or name.startswith("non_attn_cond_projection.0")
    if name.startswith("null_") or name.endswith("_codebook.embed") or name.startswith("token_embedding"):
        return synthetic_tensor(seed, key, shape, 1.0)
    if is_norm and leaf == "weight":
        return synthetic_tensor(seed, key, shape, 0.1, 1.0)
    if is_norm and leaf == "bias":
        return synthetic_tensor(seed, key, shape, 0.1)
    if leaf in ("bias", "in_proj_bias"):
        return synthetic_tensor(seed, key, shape, 0.02)
    if len(shape) == 3:
        fan_out, fan_in = shape[0] * shape[2], shape[1] * shape[2]
    else:
        fan_out, fan_in = shape[0], shape[1]
    return synthetic_tensor(seed, key, shape, float(np.sqrt(2.0 / (fan_in + fan_out))))


def synthetic_guide_state_dict(spec: GuideSpec, seed: int = 10) -> Dict[str, torch.Tensor]:
    """GuideTransformer parameters (model/guide.py) + its rotary buffer; conv weights get a gain of 2 so that 13 stacked
    LeakyReLU convolutions keep O(1) activations."""
    sd = {}
    for name, shape in guide_param_shapes(spec).items():
        t = _init_like_reference(seed, name, shape, "guide.")
        if name.startswith("pre_audio") and name.endswith("weight"):
            t = t * 2.0
        sd[name] = t
    d = spec.dim
    sd["rotary.freqs"] = 1.0 / (10000 ** (torch.arange(0, d, 2)[: d // 2].float() / d))
    return sd


def synthetic_tokenizer_state_dict(spec: TokenizerSpec, seed: int = 10) -> Dict[str, torch.Tensor]:
    return {name: _init_like_reference(seed, name, shape, "vq.") for name, shape in tokenizer_param_shapes(spec).items()}


def synthetic_frontend_state_dict(seed: int = 10, lip: bool = True, geometry=None) -> Dict[str, torch.Tensor]:
    """Parameters of the audio front end under the reference's keys (`audio_model.*`, `lip_model.*`; model/audio_frontend.py).
    Conv weights get He gain (bias-free conv + ReLU stacks: keeps the 8-layer activations O(1) on N(0,1) audio); the positional
    tables `pe` are the reference's closed form (transformer_modules.py:284-291), not random.  `geometry`
    (audio_frontend.FrontendGeometry): with fairseq's blocks the dictionary also carries the GroupNorm affine terms
    (`conv_layers.{i}.2.*`) and the lip encoder's `feature_aggregator.*` -- the key set of a real (vq-)wav2vec checkpoint's
    on-path tensors."""
    from .model.audio_frontend import STUB, Audio2LipRegressionTransformer, Wav2VecModel
    geo = geometry or STUB
    mods = {"audio_model.": Wav2VecModel(group_norm=geo.a_group_norm)}
    if lip:
        mods["lip_model."] = Audio2LipRegressionTransformer(geometry=geo)
    sd: Dict[str, torch.Tensor] = {}
    for prefix, m in mods.items():
        for name, ref in m.state_dict().items():
            key, shape = prefix + name, tuple(ref.shape)
            if name.endswith(".pe"):
                sd[key] = ref.clone()
            elif "conv_layers" in name and len(shape) == 3:
                fan_in = shape[1] * shape[2]
                sd[key] = synthetic_tensor(seed, key, shape, float(np.sqrt(2.0 / fan_in)))
            elif "conv_layers" in name and name.endswith(".weight"):      # GroupNorm scale
                sd[key] = synthetic_tensor(seed, key, shape, 0.1, 1.0)
            elif "conv_layers" in name:                                   # GroupNorm shift / aggregator conv bias
                sd[key] = synthetic_tensor(seed, key, shape, 0.1)
            else:
                sd[key] = _init_like_reference(seed, name, shape, prefix)
    return sd


def synthetic_audio(seed: int, batch: int, frames: int) -> torch.Tensor:
    """z-normalised 48 kHz stereo like data_loaders/data.py:237 hands over: N(0, 1) [batch, frames * 1600, 2]."""
    return synthetic_tensor(seed, "audio", (batch, frames * 1600, 2))


def synthetic_inputs(spec: DenoiserSpec, batch: int, frames: int, seed: int = 10,
                     steps_of_noise: int = 0) -> Dict[str, torch.Tensor]:
    """x_T, conditioning features (fed past the hoisted audio front end,
    SURVEY.md §8d "Synthetic inputs"), keyframes, mask, optional per-step noise."""
    n_tok = cond_tokens_for_frames(frames)
    out = {
        "x_T": synthetic_tensor(seed, "x_T", (batch, spec.nfeats, 1, frames)),
        "cond_embed": synthetic_tensor(seed, "cond_embed", (batch, n_tok, spec.cond_feature_dim)),
    }
    if spec.is_pose:
        nk = len(range(frames)[:: spec.keyframe_step])
        out["keyframes"] = synthetic_tensor(seed, "keyframes", (batch, nk, spec.keyframe_dim))
        out["mask"] = torch.ones(batch, 1, 1, frames, dtype=torch.bool)
    if steps_of_noise:
        out["step_noise"] = synthetic_tensor(seed, "step_noise", (steps_of_noise, batch, spec.nfeats, 1, frames))
    return out


def cond_tokens_for_frames(frames: int) -> int:
    """vq-wav2vec token count for `frames` motion frames (1600 samples @48 kHz per
    frame, 3:1 resample, conv strides 5,4,2,2,2 / kernels 10,8,4,4,4):
    600 frames -> 1998, 240 -> 798 (model/diffusion.py:136, train/train_guide.py:316)."""
    n = (frames * 1600) // 3
    for k, s in ((10, 5), (8, 4), (4, 2), (4, 2), (4, 2)):
        n = (n - k) // s + 1
    return n
