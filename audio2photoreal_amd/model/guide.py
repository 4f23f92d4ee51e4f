"""GuideTransformer: drop-in for the reference's keyframe-token predictor on the sampling path.

Mirrors `model/guide.py:26-222` at its interface -- constructor arguments, `state_dict()` key layout, `forward(tokens,
condition, cond_drop_prob)` and `generate(condition, sequence_length, layers, n_sequences, max_key_len, max_seq_len, top_p)` --
so `_replace_keyframes` (sample/generate.py:51-71) drives it like the reference's `model.transformer`.

Nothing is computed in PyTorch: the sub-modules are parameter containers; `liba2p_hip.so` hoists everything that does not
depend on the tokens (the 13-layer 1024-channel `pre_audio` conv stack the reference re-runs in each of the 80 steps, the
conditioning projections, all FiLM vectors, the cross-attention K/V of every layer) into `a2p_guide_prepare`, and runs the
whole autoregressive loop -- decoder stack with a self-attention K/V cache, softmax, sort, nucleus cut, categorical draw -- as
ONE persistent kernel launch (`a2p_guide_generate`).

`condition` is the audio FEATURE tensor [B, S, 1024] that `encode_audio` (model/guide.py:111-119) returns -- the vq-wav2vec
front end is outside this path -- or raw audio when an `audio_frontend` callable is given.
The categorical draw takes `uniforms` [sequence_length * layers, B] (default: torch.rand on the device) instead of torch's
global multinomial stream: token = first sorted index whose cumulative nucleus probability exceeds u.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

import torch
import torch.nn as nn
from torch.nn import functional as F

from .. import _lib
from .diffusion import DecoderLayerStack, RotaryEmbedding, _DecoderLayerParams


class GuideTransformer(nn.Module):
    def __init__(self, tokens: int, num_heads: int = 4, num_layers: int = 4, dim: int = 512, ff_size: int = 1024,
                 dropout: float = 0.1, activation: Callable = F.gelu, use_rotary: bool = True, cond_feature_dim: int = 1024,
                 emb_len: int = 798, num_audio_layers: int = 2, audio_frontend: Optional[Callable] = None,
                 max_batch: int = 32, max_positions: int = 96) -> None:
        super().__init__()
        if not use_rotary:
            raise NotImplementedError("absolute positional encoding is not on the accelerated path")
        if activation is not F.gelu:
            raise NotImplementedError("the reference builds the guide transformer with F.gelu")
        self.tokens, self.dim, self.num_heads, self.num_layers, self.ff_size = tokens, dim, num_heads, num_layers, ff_size
        self.cond_feature_dim, self.emb_len, self.num_audio_layers = cond_feature_dim, emb_len, num_audio_layers
        self.audio_frontend, self.max_batch, self.max_positions = audio_frontend, max_batch, max_positions
        self.token_embedding = nn.Embedding(tokens + 1, dim)      # + the sequence-start token
        self.abs_pos_encoding = nn.Identity()
        self.rotary = RotaryEmbedding(dim=dim)
        c = cond_feature_dim
        pre = []
        for _ in range(num_audio_layers):                          # model/guide.py:84-109
            for cin, cout, dl in ((c, max(256, c), 1), (max(256, c), max(256, c), 2), (max(128, c), max(128, c), 3),
                                  (max(128, c), c, 1), (c, c, 2), (c, c, 3)):
                pre += [nn.Conv1d(cin, cout, kernel_size=3, dilation=dl), nn.LeakyReLU(0.2), nn.Dropout(0.2)]
        pre += [nn.Conv1d(c, c, kernel_size=1)]
        self.pre_audio = nn.Sequential(*pre)
        self.null_cond_embed = nn.Parameter(torch.randn(1, emb_len, dim))
        self.null_cond_hidden = nn.Parameter(torch.randn(1, dim))
        self.norm_cond = nn.LayerNorm(dim)
        self.cond_projection = nn.Linear(cond_feature_dim, dim)
        self.non_attn_cond_projection = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, dim), nn.SiLU(), nn.Linear(dim, dim))
        self.seqTransDecoder = DecoderLayerStack(nn.ModuleList(
            [_DecoderLayerParams(dim, num_heads, ff_size, dropout, self.rotary, False) for _ in range(num_layers)]))
        self.final_layer = nn.Linear(dim, tokens)
        self._ctx = None
        self._sig = None
        self._prepared_for = None     # (content key of the condition tensor, cond_drop_prob) of the hoisted state ...
        self._prepared_ref = None     # ... and a strong reference that pins the keyed address (see _lib.content_key)

    # ------------------------------------------------------------------ native context
    def _signature(self):
        ps = list(self.parameters())
        return (len(ps), sum(p._version for p in ps), str(ps[0].device))

    def _ensure(self, device) -> None:
        _lib.require_gpu_tensor(self.final_layer.weight, "GuideTransformer parameters")
        lib = _lib.load()
        if self._ctx is None:
            cfg = _lib.A2PGuideConfig(self.tokens, self.dim, self.num_layers, self.num_heads, self.ff_size, self.cond_feature_dim,
                                      self.emb_len, self.num_audio_layers, self.max_batch, self.max_positions)
            ctx = C.c_void_p()
            with torch.cuda.device(device):
                _lib.check(lib.a2p_guide_create(C.byref(cfg), C.byref(ctx)), "a2p_guide_create")
            self._ctx = ctx
        sig = self._signature()
        if sig != self._sig:
            stream = _lib.current_stream(device)
            with torch.cuda.device(device):
                for k, v in self.state_dict().items():
                    if k.endswith("rotary.freqs"):
                        continue
                    t = v.detach().to(device=device, dtype=torch.float32).contiguous()
                    _lib.check(lib.a2p_guide_set_weight(self._ctx, k.encode(), _lib.ptr(t), t.numel(), stream), f"a2p_guide_set_weight({k})")
                _lib.check(lib.a2p_guide_finalize(self._ctx, stream), "a2p_guide_finalize")
            torch.cuda.current_stream(device).synchronize()
            self._sig = sig
            self.invalidate_cond()

    def __del__(self):
        if getattr(self, "_ctx", None) is not None:
            try:
                _lib.load().a2p_guide_destroy(self._ctx)
            except Exception:
                pass

    def encode_audio(self, raw_audio: torch.Tensor) -> torch.Tensor:
        if self.audio_frontend is None:
            raise _lib.A2PError("the vq-wav2vec front end is outside this path: pass the [B, S, 1024] audio features as "
                                "`condition`, or construct GuideTransformer(audio_frontend=callable)")
        return self.audio_frontend(raw_audio)

    def invalidate_cond(self) -> None:
        self._prepared_for, self._prepared_ref = None, None

    def _prepare(self, condition: torch.Tensor, cond_drop_prob: float) -> int:
        if cond_drop_prob not in (0.0, 1.0):
            raise NotImplementedError("inference uses cond_drop_prob in {0, 1}")
        _lib.require_gpu_tensor(condition, "condition")
        self._ensure(condition.device)
        # keyed on the CALLER's tensor (features or raw audio) and pinned by a strong reference: the front end and the hoisted
        # conv stack run once per clip, and a recycled address can never alias another clip's state
        key = (_lib.content_key(condition), cond_drop_prob)
        if key != self._prepared_for:
            is_feats = condition.dim() == 3 and condition.shape[-1] == self.cond_feature_dim
            feats = (condition if is_feats else self.encode_audio(condition)).to(torch.float32).contiguous()
            with _lib.on_device_of(feats):
                _lib.check(_lib.load().a2p_guide_prepare(self._ctx, _lib.ptr(feats), feats.shape[0], feats.shape[1],
                                                         int(cond_drop_prob == 1.0), _lib.current_stream(feats.device)), "a2p_guide_prepare")
            self._prepared_for, self._prepared_ref, self._prepared_batch = key, condition, feats.shape[0]
        return self._prepared_batch

    # ------------------------------------------------------------------ reference surface
    def forward(self, tokens: torch.Tensor, condition: torch.Tensor, cond_drop_prob: float = 0.0) -> torch.Tensor:
        """model/guide.py:140-173: logits [B, len, tokens] for a (causal) token prefix."""
        B = self._prepare(condition, cond_drop_prob)
        assert tokens.shape[0] == B, f"{tokens.shape[0]} token rows for {B} conditions"
        toks = tokens.to(device=condition.device, dtype=torch.int64).contiguous()
        logits = torch.empty(B, toks.shape[1], self.tokens, device=toks.device, dtype=torch.float32)
        with _lib.on_device_of(toks):
            _lib.check(_lib.load().a2p_guide_forward(self._ctx, _lib.ptr(toks), B, toks.shape[1], _lib.ptr(logits),
                                                     _lib.current_stream(toks.device)), "a2p_guide_forward")
        return logits

    def generate(self, condition: torch.Tensor, sequence_length: int, layers: int, n_sequences: int = 1, max_key_len: int = 8,
                 max_seq_len: int = 240, top_p: float = 0.94, uniforms: Optional[torch.Tensor] = None,
                 return_probs: bool = False) -> torch.Tensor:
        """model/guide.py:175-222: `sequence_length * layers` tokens per sequence, nucleus sampling; returns int64
        [n_sequences, sequence_length * layers] (the start token is not returned)."""
        assert max_key_len == int(max_seq_len / 30), "currently only running for 1fps"
        B = self._prepare(condition, 0.0)
        assert n_sequences == B, "one condition row per sequence"
        n = sequence_length * layers
        dev = condition.device
        if uniforms is None:
            uniforms = torch.rand(n, B, device=dev)
        u = uniforms.to(device=dev, dtype=torch.float32).contiguous()
        assert u.shape == (n, B), f"uniforms must be [{n}, {B}]"
        out = torch.empty(B, n, device=dev, dtype=torch.int64)
        probs = torch.empty(n, B, self.tokens, device=dev, dtype=torch.float32) if return_probs else None
        with torch.no_grad(), _lib.on_device_of(u):
            _lib.check(_lib.load().a2p_guide_generate(self._ctx, B, n, float(top_p), _lib.ptr(u), _lib.ptr(out), _lib.ptr(probs),
                                                      _lib.current_stream(dev)), "a2p_guide_generate")
        return (out, probs) if return_probs else out

    def pre_audio_features(self, n_rows: int) -> torch.Tensor:
        """test hook: output rows of the hoisted `pre_audio` stack for the last prepared condition, [B * S, C] (row stride S)."""
        import numpy as np
        host = np.empty((n_rows, self.cond_feature_dim), np.float32)
        _lib.check(_lib.load().a2p_guide_debug_read(self._ctx, b"pre_audio", host.ctypes.data_as(C.c_void_p), host.nbytes), "debug_read")
        return torch.from_numpy(host)
