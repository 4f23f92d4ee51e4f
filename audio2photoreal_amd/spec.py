"""Shape contract of the two denoisers on the hot path.

Mirrors the construction contract of the reference factory
(utils/model_util.py:49-76: face nfeats 256 / latent 512, pose nfeats 104 /
latent 256, ff 1024) and the FiLMTransformer constructor
(model/diffusion.py:83-199).  Only sizes live here; no arithmetic.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Tuple


@dataclass(frozen=True)
class DenoiserSpec:
    data_format: str            # "face" | "pose"
    nfeats: int                 # motion feature channels C
    latent_dim: int             # d_model
    num_layers: int
    num_heads: int
    ff_size: int = 1024
    cond_feature_dim: int = 1024  # width of the audio conditioning features
    max_seq_length: int = 600     # frames
    emb_len: int = 1998           # model/diffusion.py:136 ("hardcoded for now")
    keyframe_dim: int = 104
    keyframe_step: int = 30       # model/diffusion.py:147

    @property
    def is_pose(self) -> bool:
        return self.data_format == "pose"

    @property
    def head_dim(self) -> int:
        return self.latent_dim // self.num_heads

    @property
    def max_keyframes(self) -> int:
        # len(range(seq_len)[::step])  (model/diffusion.py:228)
        return len(range(self.max_seq_length)[:: self.keyframe_step])

    @property
    def num_films(self) -> int:
        return 4 if self.is_pose else 3


def face_spec(num_layers: int = 8, num_heads: int = 8, **kw) -> DenoiserSpec:
    """README.md:295 face model: 8 layers / 8 heads, cond = 1024 audio + 1014 lip."""
    return DenoiserSpec("face", 256, 512, num_layers, num_heads,
                        cond_feature_dim=1024 + 1014, **kw)


def pose_spec(num_layers: int = 6, num_heads: int = 8, **kw) -> DenoiserSpec:
    """README.md:322 body model: 6 layers / 8 heads, add_frame_cond=1."""
    return DenoiserSpec("pose", 104, 256, num_layers, num_heads,
                        cond_feature_dim=1024, **kw)


def param_shapes(spec: DenoiserSpec) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict keys/shapes of the hot-path parameters.

    Same names as the reference FiLMTransformer so checkpoints load unchanged.
    `audio_model.*` / `lip_model.*` (conditioning producers, out of scope) and the
    duplicated `*.rotary.freqs` buffers are not listed here.
    """
    d, C, ff = spec.latent_dim, spec.nfeats, spec.ff_size
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["null_cond_embed"] = (1, spec.emb_len, d)
    s["null_cond_hidden"] = (1, d)
    if spec.is_pose:
        s["null_pose_embed"] = (1, spec.max_keyframes, d)
    s["time_mlp.1.weight"] = (4 * d, d)
    s["time_mlp.1.bias"] = (4 * d,)
    s["to_time_cond.0.weight"] = (d, 4 * d)
    s["to_time_cond.0.bias"] = (d,)
    s["to_time_tokens.0.weight"] = (2 * d, 4 * d)
    s["to_time_tokens.0.bias"] = (2 * d,)
    s["norm_cond.weight"] = (d,)
    s["norm_cond.bias"] = (d,)
    s["input_projection.weight"] = (d, C)
    s["input_projection.bias"] = (d,)
    if spec.is_pose:
        s["frame_cond_projection.weight"] = (d, spec.keyframe_dim)
        s["frame_cond_projection.bias"] = (d,)
        s["frame_norm_cond.weight"] = (d,)
        s["frame_norm_cond.bias"] = (d,)
        chans = [(C, max(256, C)), (max(256, C), C), (C, C), (C, C), (C, C), (C, C)]
        for i, (ci, co) in enumerate(chans):
            s[f"post_pose_layers.{i}.weight"] = (co, ci, 3)
            s[f"post_pose_layers.{i}.bias"] = (co,)
        s["final_conv.weight"] = (C, C, 1)
        s["final_conv.bias"] = (C,)
    else:
        for i in range(2):
            p = f"cond_encoder.{i}."
            s[p + "self_attn.in_proj_weight"] = (3 * d, d)
            s[p + "self_attn.in_proj_bias"] = (3 * d,)
            s[p + "self_attn.out_proj.weight"] = (d, d)
            s[p + "self_attn.out_proj.bias"] = (d,)
            s[p + "linear1.weight"] = (ff, d)
            s[p + "linear1.bias"] = (ff,)
            s[p + "linear2.weight"] = (d, ff)
            s[p + "linear2.bias"] = (d,)
            for n in ("norm1", "norm2"):
                s[p + n + ".weight"] = (d,)
                s[p + n + ".bias"] = (d,)
    s["cond_projection.weight"] = (d, spec.cond_feature_dim)
    s["cond_projection.bias"] = (d,)
    s["non_attn_cond_projection.0.weight"] = (d,)
    s["non_attn_cond_projection.0.bias"] = (d,)
    s["non_attn_cond_projection.1.weight"] = (d, d)
    s["non_attn_cond_projection.1.bias"] = (d,)
    s["non_attn_cond_projection.3.weight"] = (d, d)
    s["non_attn_cond_projection.3.bias"] = (d,)
    attns = ["self_attn", "multihead_attn"] + (["multihead_attn2"] if spec.is_pose else [])
    norms = ["norm1", "norm2", "norm3"] + (["norm2a"] if spec.is_pose else [])
    films = ["film1", "film2", "film3"] + (["film2a"] if spec.is_pose else [])
    for l in range(spec.num_layers):
        p = f"seqTransDecoder.stack.{l}."
        for a in attns:
            s[p + a + ".in_proj_weight"] = (3 * d, d)
            s[p + a + ".in_proj_bias"] = (3 * d,)
            s[p + a + ".out_proj.weight"] = (d, d)
            s[p + a + ".out_proj.bias"] = (d,)
        s[p + "linear1.weight"] = (ff, d)
        s[p + "linear1.bias"] = (ff,)
        s[p + "linear2.weight"] = (d, ff)
        s[p + "linear2.bias"] = (d,)
        for n in norms:
            s[p + n + ".weight"] = (d,)
            s[p + n + ".bias"] = (d,)
        for f in films:
            s[p + f + ".block.1.weight"] = (2 * d, d)
            s[p + f + ".block.1.bias"] = (2 * d,)
    s["final_layer.weight"] = (C, d)
    s["final_layer.bias"] = (C,)
    return s


def param_count(spec: DenoiserSpec) -> int:
    n = 0
    for shp in param_shapes(spec).values():
        k = 1
        for v in shp:
            k *= v
        n += k
    return n
