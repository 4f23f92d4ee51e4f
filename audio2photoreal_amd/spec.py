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


# ----------------------------------------------------------------------------------------------------------
# the guide transformer that predicts the body model's keyframe tokens, and the residual-VQ
# tokenizer that decodes them
# ----------------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class GuideSpec:
    """GuideTransformer constructor arguments (model/guide.py:26-39) at the sizes the reference trains and loads
    (README.md:350-363 `--layers 6 --dim 64`, train/train_guide.py:314-319, model/diffusion.py:253-259)."""
    tokens: int = 1024            # tokenizer.n_clusters; id `tokens` is the sequence-start token
    num_heads: int = 4
    num_layers: int = 6
    dim: int = 64
    ff_size: int = 1024
    cond_feature_dim: int = 1024
    emb_len: int = 1998
    num_audio_layers: int = 2

    @property
    def audio_conv_dilations(self) -> Tuple[int, ...]:
        return (1, 2, 3, 1, 2, 3) * self.num_audio_layers        # model/guide.py:84-109, kernel 3, no padding

    def cond_tokens_after_conv(self, n_tokens: int) -> int:
        return n_tokens - 2 * sum(self.audio_conv_dilations)


@dataclass(frozen=True)
class TokenizerSpec:
    """TemporalVertexCodec as built by setup_tokenizer (model/vqvae.py:18-34) from README.md:344
    (`--code_dim 1024 --output_emb_width 64 --depth 4`, pose: nb_joints = 104)."""
    n_vertices: int = 104
    latent_dim: int = 64
    categories: int = 1024
    residual_depth: int = 4


def guide_param_shapes(spec: GuideSpec) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict keys/shapes of GuideTransformer without the audio front end (`audio_model.*`) and the duplicated
    `*.rotary.freqs` buffers (model/guide.py:40-83,111-119)."""
    d, ff, c = spec.dim, spec.ff_size, spec.cond_feature_dim
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["token_embedding.weight"] = (spec.tokens + 1, d)
    s["null_cond_embed"] = (1, spec.emb_len, d)
    s["null_cond_hidden"] = (1, d)
    s["norm_cond.weight"] = (d,)
    s["norm_cond.bias"] = (d,)
    s["cond_projection.weight"] = (d, c)
    s["cond_projection.bias"] = (d,)
    s["non_attn_cond_projection.0.weight"] = (d,)
    s["non_attn_cond_projection.0.bias"] = (d,)
    s["non_attn_cond_projection.1.weight"] = (d, d)
    s["non_attn_cond_projection.1.bias"] = (d,)
    s["non_attn_cond_projection.3.weight"] = (d, d)
    s["non_attn_cond_projection.3.bias"] = (d,)
    n = 0
    for _ in spec.audio_conv_dilations:                      # Conv1d, LeakyReLU, Dropout triples: indices 0, 3, 6, ...
        s[f"pre_audio.{n}.weight"] = (c, c, 3)            # max(256, c) = max(128, c) = c for c = 1024
        s[f"pre_audio.{n}.bias"] = (c,)
        n += 3
    s[f"pre_audio.{n}.weight"] = (c, c, 1)
    s[f"pre_audio.{n}.bias"] = (c,)
    for l in range(spec.num_layers):
        p = f"seqTransDecoder.stack.{l}."
        for a in ("self_attn", "multihead_attn"):
            s[p + a + ".in_proj_weight"] = (3 * d, d)
            s[p + a + ".in_proj_bias"] = (3 * d,)
            s[p + a + ".out_proj.weight"] = (d, d)
            s[p + a + ".out_proj.bias"] = (d,)
        s[p + "linear1.weight"] = (ff, d)
        s[p + "linear1.bias"] = (ff,)
        s[p + "linear2.weight"] = (d, ff)
        s[p + "linear2.bias"] = (d,)
        for nm in ("norm1", "norm2", "norm3"):
            s[p + nm + ".weight"] = (d,)
            s[p + nm + ".bias"] = (d,)
        for f in ("film1", "film2", "film3"):
            s[p + f + ".block.1.weight"] = (2 * d, d)
            s[p + f + ".block.1.bias"] = (2 * d,)
    s["final_layer.weight"] = (spec.tokens, d)
    s["final_layer.bias"] = (spec.tokens,)
    return s


def tokenizer_param_shapes(spec: TokenizerSpec) -> "OrderedDict[str, Tuple[int, ...]]":
    """The decode-side parameters of TemporalVertexCodec (model/vqvae.py:432-463, 381-392): the residual codebooks and the
    causal dilated Conv1d decoder.  (Encoder, EMA statistics and `project_mean_shape` are not used by `decode`.)"""
    e = spec.latent_dim
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for i in range(spec.residual_depth):
        s[f"quantizer.layers.{i}._codebook.embed"] = (spec.categories, e)
    for i in (0, 2, 4, 6):
        s[f"decoder.dec.{i}.weight"] = (e, e, 2)
        s[f"decoder.dec.{i}.bias"] = (e,)
    s["decoder.dec.8.weight"] = (spec.n_vertices, e, 1)
    s["decoder.dec.8.bias"] = (spec.n_vertices,)
    return s
