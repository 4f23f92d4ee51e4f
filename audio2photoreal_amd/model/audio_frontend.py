"""Audio front end of the denoisers on the GPU: what the reference's `FiLMTransformer.forward` derives from
`y["audio"]` before anything else -- `encode_audio` (model/diffusion.py:285-293) and, for the face model, `encode_lip`
(:295-313) -- in every denoising step and guidance pass.  Here it runs once per clip, inside `FiLMTransformer.prepare`.

The torch modules below are parameter CONTAINERS with the reference's `state_dict()` key layout (nothing is computed in
PyTorch); the arithmetic is the `a2p_frontend_*` entry points of liba2p_hip*.so (csrc/a2p_frontend.h: fp32, or -- following the
owner model's precision -- the conv stacks' GEMMs on 16-bit operands).

  audio_model   setup_lip_regressor()'s vq-wav2vec model (model/utils.py:18-26).  Only its conv feature extractor is on the
                path (`audio_model.feature_extractor(a)`, model/diffusion.py:290-291); keys follow fairseq's
                ConvFeatureExtractionModel: `audio_model.feature_extractor.conv_layers.{i}.0.weight`.
  lip_model     Audio2LipRegressionTransformer (model/diffusion.py:37-79): Wav2VecEncoder (audio_encoder.py:24-46) +
                RegressionTransformer (transformer_modules.py:560-627) + Linear(512, 1014).

fairseq and torchaudio are absent offline: the conv stacks implement the STUB geometry of SURVEY.md Appendix A (8 bias-free
Conv1d + ReLU layers, identity aggregator) -- the published layer shapes WITHOUT fairseq's per-layer GroupNorm, log
compression and the lip encoder's ConvAggregator -- and the resampler is torchaudio's documented windowed-sinc kernel ("sinc",
default) or the 3:1 decimation of the golden generator's stub ("decimate").  Everything that IS in the reference tree -- the
regression transformer, chunking, interpolation, concat -- is pinned by reference-generated goldens
(tests/golden/golden_frontend_v1.npz).  A real fairseq checkpoint carries tensors this geometry does not consume; the ones
that sit on the conditioning path are REFUSED (`check_keys`, `a2p_frontend_set_weight`), never skipped.
"""
from __future__ import annotations

import ast
import ctypes as C
import dataclasses
import math
from typing import Optional

import torch
import torch.nn as nn

from .. import _lib

CONV_GEOMETRY = ((10, 5), (8, 4), (4, 2), (4, 2), (4, 2), (1, 1), (1, 1), (1, 1))   # (kernel, stride), 512 channels

# Sub-modules of the two wav2vec models whose tensors the reference's conditioning path READS (model/diffusion.py:285-313,
# model/modules/audio_encoder.py:43-44): a checkpoint tensor under one of these that the native front end does not implement
# changes the features, so it is an error.  Everything else under audio_model.* / lip_model.* (vector quantiser, the
# vq-wav2vec aggregator that encode_audio never calls, the prediction heads, torchaudio's resampling kernel buffer -- recomputed
# here) is not read by the reference on this path either and is skipped.
ON_PATH_PREFIXES = ("audio_model.feature_extractor.", "lip_model.audio_encoder.wav2vec_model.feature_extractor.",
                    "lip_model.audio_encoder.wav2vec_model.feature_aggregator.", "lip_model.regression_model.",
                    "lip_model.project_output.")


# ----------------------------------------------------------------------------- geometry of the two fairseq models
@dataclasses.dataclass(frozen=True)
class FrontendGeometry:
    """Block options of the two wav2vec models (include/a2p_hip.h a2p_frontend_config a_* / l_* / agg_*), named after the fairseq
    arguments they come from (fairseq 0.12 models/wav2vec/wav2vec.py -- absent offline: PARITY UNPINNED).  `a_*`: the vq-wav2vec
    model behind `audio_model.feature_extractor`; `l_*` / `agg_*`: the wav2vec-large model behind the lip encoder."""
    a_group_norm: bool = False          # Conv1d -> Dropout -> Fp32GroupNorm(1, C, affine) -> activation  (conv_layers.{i}.2.*)
    a_activation: str = "relu"          # --activation
    a_log_compression: bool = False     # --log-compression
    a_skip: bool = False                # --skip-connections-feat
    a_residual_scale: float = 0.5       # --residual-scale
    l_group_norm: bool = False
    l_activation: str = "relu"
    l_log_compression: bool = False
    l_skip: bool = False
    l_residual_scale: float = 0.5
    l_layers: int = 8                   # --conv-feature-layers: 8 = the vq-wav2vec list, 7 = wav2vec-large's (one (512, 1, 1) less)
    agg_layers: int = 0                 # --conv-aggregator-layers [(512, 2, 1) ... (512, n + 1, 1)]; 0 = identity (stub)
    agg_skip: bool = False              # --skip-connections-agg
    agg_residual_scale: float = 0.5
    agg_conv_bias: bool = False         # not --no-conv-bias
    agg_zero_pad: bool = False          # --agg-zero-pad (default: ReplicationPad1d)
    agg_activation: str = "relu"

    @property
    def is_stub(self) -> bool:
        return self == STUB

    @staticmethod
    def from_fairseq_args(audio_args=None, lip_args=None) -> "FrontendGeometry":
        """From the `args` / `cfg.model` namespaces (or dicts) stored in vq-wav2vec.pt (`audio_args`) and wav2vec_large.pt (`lip_args`)."""
        def get(a, k, d):
            return d if a is None else (a.get(k, d) if isinstance(a, dict) else getattr(a, k, d))

        def nlayers(a, k, d):
            v = get(a, k, None)
            return d if v is None else len(ast.literal_eval(v) if isinstance(v, str) else v)   # checkpoint metadata: never eval()
        kw = {}
        if audio_args is not None:
            kw.update(a_group_norm=True, a_activation=get(audio_args, "activation", "relu"), a_log_compression=bool(get(audio_args, "log_compression", False)),
                      a_skip=bool(get(audio_args, "skip_connections_feat", False)), a_residual_scale=float(get(audio_args, "residual_scale", 0.5)))
            assert nlayers(audio_args, "conv_feature_layers", 8) == 8, "audio_model: the 8-layer vq-wav2vec feature extractor is the one on the path"
        if lip_args is not None:
            kw.update(l_group_norm=True, l_activation=get(lip_args, "activation", "relu"), l_log_compression=bool(get(lip_args, "log_compression", False)),
                      l_skip=bool(get(lip_args, "skip_connections_feat", False)), l_residual_scale=float(get(lip_args, "residual_scale", 0.5)),
                      l_layers=nlayers(lip_args, "conv_feature_layers", 8), agg_layers=nlayers(lip_args, "conv_aggregator_layers", 0),
                      agg_skip=bool(get(lip_args, "skip_connections_agg", False)), agg_residual_scale=float(get(lip_args, "residual_scale", 0.5)),
                      agg_conv_bias=not bool(get(lip_args, "no_conv_bias", False)), agg_zero_pad=bool(get(lip_args, "agg_zero_pad", False)),
                      agg_activation=get(lip_args, "activation", "relu"))
        return FrontendGeometry(**kw)


STUB = FrontendGeometry()
# The training commands of fairseq's examples/wav2vec/README.md (as published; a checkpoint's own `args` are authoritative:
# FrontendGeometry.from_fairseq_args): vq-wav2vec -- 8 feature layers, --activation gelu, --log-compression; wav2vec-large -- 7 feature
# layers, 12 aggregator layers of kernel 2..13, --skip-connections-agg --residual-scale 0.5 --log-compression, conv bias on
FAIRSEQ = FrontendGeometry(a_group_norm=True, a_activation="gelu", a_log_compression=True,
                           l_group_norm=True, l_activation="relu", l_log_compression=True, l_layers=7,
                           agg_layers=12, agg_skip=True, agg_residual_scale=0.5, agg_conv_bias=True)
_ACT = {"relu": 0, "gelu": 1}


# ----------------------------------------------------------------------------- parameter containers (reference key layout)
class ConvFeatureExtractor(nn.Module):
    """fairseq ConvFeatureExtractionModel: conv_layers.{i} = Sequential(Conv1d(bias=False), Dropout, Fp32GroupNorm(1, C), act) --
    parameters `.0.weight` and, with `group_norm`, `.2.weight` / `.2.bias`."""

    def __init__(self, dim: int = 512, layers: int = 8, group_norm: bool = False):
        super().__init__()
        blocks, cin = [], 1
        for k, s in CONV_GEOMETRY[:layers]:
            mods = [nn.Conv1d(cin, dim, k, stride=s, bias=False)]
            if group_norm:
                mods += [nn.Dropout(0.0), nn.GroupNorm(1, dim)]
            blocks.append(nn.Sequential(*mods))
            cin = dim
        self.conv_layers = nn.ModuleList(blocks)


class ConvAggregator(nn.Module):
    """fairseq ConvAggregator: conv_layers.{j} = Sequential(pad, Conv1d(C, C, j + 2), Dropout, Fp32GroupNorm(1, C), act) --
    parameters `.1.weight` [, `.1.bias`], `.3.weight`, `.3.bias` (residual_proj holds no modules: all layers are C -> C)."""

    def __init__(self, dim: int = 512, layers: int = 12, conv_bias: bool = True):
        super().__init__()
        self.conv_layers = nn.ModuleList([nn.Sequential(nn.Identity(), nn.Conv1d(dim, dim, j + 2, bias=conv_bias), nn.Dropout(0.0), nn.GroupNorm(1, dim))
                                          for j in range(layers)])


class Wav2VecModel(nn.Module):
    def __init__(self, layers: int = 8, group_norm: bool = False, agg_layers: int = 0, agg_conv_bias: bool = True):
        super().__init__()
        self.feature_extractor = ConvFeatureExtractor(layers=layers, group_norm=group_norm)
        if agg_layers:
            self.feature_aggregator = ConvAggregator(layers=agg_layers, conv_bias=agg_conv_bias)


class Wav2VecEncoder(nn.Module):
    """model/modules/audio_encoder.py:24-46."""

    def __init__(self, geometry: "FrontendGeometry" = None):
        super().__init__()
        g = geometry or STUB
        self.wav2vec_model = Wav2VecModel(layers=g.l_layers, group_norm=g.l_group_norm, agg_layers=g.agg_layers, agg_conv_bias=g.agg_conv_bias)


class PositionalEncoding(nn.Module):
    """transformer_modules.py:281-302; `pe` is a registered buffer of the reference, so it is part of the state_dict."""

    def __init__(self, d_model: int, max_len: int = 1024):
        super().__init__()
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)


class _SelfAttention(nn.Module):
    def __init__(self, d, h):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d, h, batch_first=True)


class _CrossAttention(nn.Module):
    def __init__(self, d, h):
        super().__init__()
        self.cross_attn = nn.MultiheadAttention(d, h, batch_first=True, kdim=d, vdim=d)


class _Feedforward(nn.Module):
    def __init__(self, d, ff):
        super().__init__()
        self.ff = nn.Sequential(nn.Linear(d, ff), nn.ReLU(), nn.Dropout(0.1), nn.Linear(ff, d), nn.Dropout(0.1))


class _EncoderLayer(nn.Module):
    def __init__(self, d, h, ff):
        super().__init__()
        self.norm1, self.self_attn, self.norm2, self.feedforward = nn.LayerNorm(d), _SelfAttention(d, h), nn.LayerNorm(d), _Feedforward(d, ff)


class _DecoderLayer(nn.Module):
    def __init__(self, d, h, ff):
        super().__init__()
        self.norm1, self.self_attn = nn.LayerNorm(d), _SelfAttention(d, h)
        self.norm2, self.cross_attn = nn.LayerNorm(d), _CrossAttention(d, h)
        self.norm3, self.feedforward = nn.LayerNorm(d), _Feedforward(d, ff)


class RegressionTransformer(nn.Module):
    def __init__(self, enc: int = 2, dec: int = 4, d: int = 512, h: int = 4, ff: int = 1024):
        super().__init__()
        self.cond_positional_encoding = PositionalEncoding(d)
        self.target_positional_encoding = PositionalEncoding(d)
        self.transformer_encoder = nn.ModuleList([_EncoderLayer(d, h, ff) for _ in range(enc)])
        self.transformer_decoder = nn.ModuleList([_DecoderLayer(d, h, ff) for _ in range(dec)])


class Audio2LipRegressionTransformer(nn.Module):
    """model/diffusion.py:37-79 (parameters only)."""

    def __init__(self, n_vertices: int = 338, geometry: "FrontendGeometry" = None):
        super().__init__()
        self.n_vertices = n_vertices
        self.audio_encoder = Wav2VecEncoder(geometry)
        self.regression_model = RegressionTransformer()
        self.project_output = nn.Linear(512, n_vertices * 3)


# ----------------------------------------------------------------------------- native front end
class NativeAudioFrontend:
    """Drives `a2p_frontend_*` with the parameters of an owner module's `audio_model` (and `lip_model`, if present).

    `encode_audio(audio)` = FiLMTransformer.encode_audio (also what GuideTransformer.encode_audio computes, model/guide.py:111-119);
    `__call__(audio)` = encode_audio followed by encode_lip when the owner has a lip model: the denoiser's `cond_embed`."""

    def __init__(self, owner: nn.Module, resample: str = "sinc", max_batch: int = 32, max_frames: int = 600,
                 precision: Optional[str] = None, geometry: "FrontendGeometry" = None):
        """`precision`: "fp32" = exact-fp32 MFMA everywhere (parity mode); "fp16" / "bf16" = the conv feature extractors' GEMMs (99 % of
        the front end's FLOPs) on 16-bit operands with fp32 accumulation; None = follow `owner.precision` at first use."""
        assert resample in ("sinc", "decimate")
        assert precision in (None, "fp32", "fp16", "f16", "bf16")
        self.owner, self.resample, self.max_batch, self.max_frames = owner, resample, max_batch, max_frames
        self.precision = precision
        self.geometry = geometry or STUB
        self._ctx, self._sig, self._ctx_lib = None, None, None

    def check_keys(self, unconsumed) -> None:
        """`unconsumed`: checkpoint keys under audio_model.* / lip_model.* that the parameter containers do not hold.  Raises for
        the ones on the conditioning path (ON_PATH_PREFIXES): fairseq's real ConvFeatureExtractionModel block is Conv1d -> Dropout ->
        Fp32GroupNorm(1, 512) -> activation (+ log compression after the stack), and Wav2VecEncoder.forward also runs the 12-layer
        ConvAggregator (audio_encoder.py:44).  With the stub geometry none of that exists; build the model with
        `audio_geometry=FAIRSEQ` (or FrontendGeometry.from_fairseq_args(...)) and the containers hold -- and the library consumes --
        those tensors."""
        has_lip = self.has_lip
        bad = sorted(k for k in unconsumed if k.startswith(ON_PATH_PREFIXES) and (has_lip or not k.startswith("lip_model.")))
        if bad:
            raise _lib.A2PError(
                f"the native audio front end does not implement {len(bad)} checkpoint tensor(s) that sit on the conditioning path, e.g. "
                f"{bad[:3]} (GroupNorm affine terms / feature aggregator of a real fairseq wav2vec checkpoint): loading would silently "
                f"change the features.  Construct the model with audio_geometry=audio_frontend.FAIRSEQ (fairseq's published blocks; "
                f"FrontendGeometry.from_fairseq_args reads a checkpoint's own options), feed y['cond_embed'] computed by the reference's "
                f"encode_audio / encode_lip (audio_frontend=None), or use weights exported for the stub geometry.")

    def _params(self):
        out = {}
        for prefix in ("audio_model", "lip_model"):
            m = getattr(self.owner, prefix, None)
            if isinstance(m, nn.Module):
                out.update({f"{prefix}.{k}": v for k, v in m.state_dict().items()})
        return out

    @property
    def has_lip(self) -> bool:
        return isinstance(getattr(self.owner, "lip_model", None), nn.Module)

    def _precision(self) -> str:
        return self.precision or getattr(self.owner, "precision", "fp32")

    def _lib(self):
        return _lib.load(half=self._precision() in ("fp16", "f16"))

    def _ensure(self, device) -> None:
        prec = self._precision()
        lib = self._lib()
        params = self._params()
        sig = (str(device), prec, self.geometry, _lib.content_key(*params.values()))
        if self._ctx is not None and sig == self._sig:
            return
        self.release()
        cfg = _lib.A2PFrontendConfig(conv_dim=512, resample=int(self.resample == "sinc"), lip=int(self.has_lip), d_model=512, num_heads=4,
                                     ff_size=1024, enc_layers=2, dec_layers=4, lip_out=1014, lip_pad=320, chunk_frames=120,
                                     samples_per_frame=1600, max_batch=self.max_batch, max_frames=self.max_frames,
                                     conv_16bit=int(prec != "fp32"))
        g = self.geometry
        for k, v in dataclasses.asdict(g).items():
            setattr(cfg, k, _ACT[v] if k.endswith("activation") else (float(v) if k.endswith("residual_scale") else int(v)))
        ctx = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.a2p_frontend_create(C.byref(cfg), C.byref(ctx)), "a2p_frontend_create")
            stream = _lib.current_stream(device)
            keep = []
            for name, t in params.items():
                t = t.detach().to(device=device, dtype=torch.float32).contiguous()
                keep.append(t)
                _lib.check(lib.a2p_frontend_set_weight(ctx, name.encode(), _lib.ptr(t), t.numel(), stream), f"a2p_frontend_set_weight({name})")
            _lib.check(lib.a2p_frontend_finalize(ctx, stream), "a2p_frontend_finalize")
        self._ctx, self._sig, self._keep, self._ctx_lib = ctx, sig, list(params.values()), lib

    def release(self) -> None:
        if self._ctx is not None:
            self._ctx_lib.a2p_frontend_destroy(self._ctx)   # the library build that created it
            self._ctx = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    @staticmethod
    def n_tokens(samples48: int) -> int:   # (the (1, 1) layers do not change the length: the same count for 7 and 8 layers)
        n = (samples48 + 2) // 3
        for k, s in CONV_GEOMETRY:
            n = (n - k) // s + 1
        return n

    def encode_audio(self, audio: torch.Tensor) -> torch.Tensor:
        """audio fp32 [B, samples, 2] (48 kHz, ch0 = self, ch1 = partner) -> [B, n_tokens, 1024]."""
        _lib.require_gpu_tensor(audio, "audio")
        assert audio.dim() == 3 and audio.shape[-1] == 2, f"audio must be [B, samples, 2], got {tuple(audio.shape)}"
        self._ensure(audio.device)
        a = audio.to(torch.float32).contiguous()
        B, L = a.shape[0], a.shape[1]
        S = self.n_tokens(L)
        out = torch.empty(B, S, 1024, device=a.device, dtype=torch.float32)
        with _lib.on_device_of(a):
            _lib.check(self._ctx_lib.a2p_frontend_encode_audio(self._ctx, _lib.ptr(a), B, L, _lib.ptr(out), S, _lib.current_stream(a.device)),
                       "a2p_frontend_encode_audio")
        return out

    def encode_lip(self, audio: torch.Tensor, cond_embed: torch.Tensor) -> torch.Tensor:
        """[B, S, Ca] -> [B, S, Ca + 1014] (model/diffusion.py:295-313)."""
        self._ensure(audio.device)
        a, ce = audio.to(torch.float32).contiguous(), cond_embed.to(torch.float32).contiguous()
        B, L, S, Ca = a.shape[0], a.shape[1], ce.shape[1], ce.shape[2]
        out = torch.empty(B, S, Ca + 1014, device=a.device, dtype=torch.float32)
        with _lib.on_device_of(a):
            _lib.check(self._ctx_lib.a2p_frontend_encode_lip(self._ctx, _lib.ptr(a), B, L, _lib.ptr(ce), S, Ca, _lib.ptr(out),
                                                           _lib.current_stream(a.device)), "a2p_frontend_encode_lip")
        return out

    def __call__(self, audio: torch.Tensor) -> torch.Tensor:
        ce = self.encode_audio(audio)
        return self.encode_lip(audio, ce) if self.has_lip else ce
