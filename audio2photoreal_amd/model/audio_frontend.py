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

import ctypes as C
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


# ----------------------------------------------------------------------------- parameter containers (reference key layout)
class ConvFeatureExtractor(nn.Module):
    def __init__(self, dim: int = 512):
        super().__init__()
        layers, cin = [], 1
        for k, s in CONV_GEOMETRY:
            layers.append(nn.Sequential(nn.Conv1d(cin, dim, k, stride=s, bias=False)))
            cin = dim
        self.conv_layers = nn.ModuleList(layers)


class Wav2VecModel(nn.Module):
    def __init__(self):
        super().__init__()
        self.feature_extractor = ConvFeatureExtractor()


class Wav2VecEncoder(nn.Module):
    """model/modules/audio_encoder.py:24-46."""

    def __init__(self):
        super().__init__()
        self.wav2vec_model = Wav2VecModel()


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

    def __init__(self, n_vertices: int = 338):
        super().__init__()
        self.n_vertices = n_vertices
        self.audio_encoder = Wav2VecEncoder()
        self.regression_model = RegressionTransformer()
        self.project_output = nn.Linear(512, n_vertices * 3)


# ----------------------------------------------------------------------------- native front end
class NativeAudioFrontend:
    """Drives `a2p_frontend_*` with the parameters of an owner module's `audio_model` (and `lip_model`, if present).

    `encode_audio(audio)` = FiLMTransformer.encode_audio (also what GuideTransformer.encode_audio computes, model/guide.py:111-119);
    `__call__(audio)` = encode_audio followed by encode_lip when the owner has a lip model: the denoiser's `cond_embed`."""

    def __init__(self, owner: nn.Module, resample: str = "sinc", max_batch: int = 32, max_frames: int = 600,
                 precision: Optional[str] = None):
        """`precision`: "fp32" = exact-fp32 MFMA everywhere (parity mode); "fp16" / "bf16" = the conv feature extractors' GEMMs (99 % of
        the front end's FLOPs) on 16-bit operands with fp32 accumulation; None = follow `owner.precision` at first use."""
        assert resample in ("sinc", "decimate")
        assert precision in (None, "fp32", "fp16", "f16", "bf16")
        self.owner, self.resample, self.max_batch, self.max_frames = owner, resample, max_batch, max_frames
        self.precision = precision
        self._ctx, self._sig, self._ctx_lib = None, None, None

    def check_keys(self, unconsumed) -> None:
        """`unconsumed`: checkpoint keys under audio_model.* / lip_model.* that the parameter containers do not hold.  Raises for
        the ones on the conditioning path (ON_PATH_PREFIXES): fairseq's real ConvFeatureExtractionModel block is Conv1d -> Dropout ->
        Fp32GroupNorm(1, 512) -> activation (+ log compression after the stack), and Wav2VecEncoder.forward also runs the 12-layer
        ConvAggregator (audio_encoder.py:44); the native front end implements the stub geometry only."""
        has_lip = self.has_lip
        bad = sorted(k for k in unconsumed if k.startswith(ON_PATH_PREFIXES) and (has_lip or not k.startswith("lip_model.")))
        if bad:
            raise _lib.A2PError(
                f"the native audio front end does not implement {len(bad)} checkpoint tensor(s) that sit on the conditioning path, e.g. "
                f"{bad[:3]} (GroupNorm affine terms / feature aggregator of a real fairseq wav2vec checkpoint): loading would silently "
                f"change the features.  Feed y['cond_embed'] computed by the reference's encode_audio / encode_lip instead "
                f"(audio_frontend=None), or use weights exported for the stub geometry.")

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
        sig = (str(device), prec, _lib.content_key(*params.values()))
        if self._ctx is not None and sig == self._sig:
            return
        self.release()
        cfg = _lib.A2PFrontendConfig(conv_dim=512, resample=int(self.resample == "sinc"), lip=int(self.has_lip), d_model=512, num_heads=4,
                                     ff_size=1024, enc_layers=2, dec_layers=4, lip_out=1014, lip_pad=320, chunk_frames=120,
                                     samples_per_frame=1600, max_batch=self.max_batch, max_frames=self.max_frames,
                                     conv_16bit=int(prec != "fp32"))
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
    def n_tokens(samples48: int) -> int:
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
