"""ORACLE-SIDE TOOL -- test infrastructure, NOT product code.

A CPU *model of the GPU's 16-bit throughput modes*: the oracle's denoiser (oracle/a2p_oracle.py, same reference line
citations) with an explicit rounding hook at every place where the HIP path stores or stages a 16-bit operand
(csrc/kernels_chain.h, kernels_attn.h, a2p_lib_run.h).  Accumulation, LayerNorm statistics, the residual stream, FiLM, the
time path and the sampler stay fp32 exactly as on the GPU.  It answers "which operand class carries how much of the error of
the loop's return value" without GPU time (tests/tools/error_budget.py -> profiles/r03_error_budget.json) and predicts the
error of a candidate operand format before its kernel exists.

Sites (each is one class of rounded operand; `Rounding.modes[site]` picks its format, default = Rounding.default):
  in.a in.w            x_t rows packed for input_projection / its weight
  qkv.a qkv.w          LN(norm1) (+rotary) panel -> self-attention Q|K|V projections / in_proj weight
  self.q self.k self.v stored Q, K, V^T of the self attention
  self.p               softmax numerators P of the self attention
  self.o               attention output rows (operand of out_proj)
  oself.w              self_attn.out_proj weight
  qc.a qc.w            LN(norm2) + rotary panel -> cross-attention query projection / weight
  cross.q              stored query of the cross attention
  cross.kv             cached audio K / V^T (the STORAGE rounding of the cache; the conditioning path that produces them is `cond.*`)
  cross.tail           the two per-step time-token K/V rows patched into the last key tile
  cross.p cross.o      as self.*
  ocross.w             multihead_attn.out_proj weight
  ff1.a ff1.w          LN(norm3) panel / linear1 weight
  ff2.a ff2.w          GELU hidden / linear2 weight
  fin.a fin.w          final_layer operand rows / weight
  cond.a cond.w        every operand of the hoisted conditioning path (cond_projection, face cond_encoder, K/V projections)
  pose only: qc2.a qc2.w cross2.q cross2.kv cross2.p cross2.o ocross2.w (keyframe cross attention), tail.a tail.w (conv tail)

Formats: "fp32" (no rounding), "fp16", "bf16", "fp16x2" (hi + lo IEEE-half pair: what a split-operand MFMA pair sees),
"fp16+e4m3" / "fp16+e5m2" (IEEE-half hi + an 8-bit float lo scaled per tensor by a power of two: the compensated scheme).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import a2p_oracle as O

Tensor = torch.Tensor


def _round(x: Tensor, fmt: str) -> Tensor:
    if fmt == "fp32":
        return x
    if fmt == "fp16":
        return x.half().float()
    if fmt == "bf16":
        return x.bfloat16().float()
    if fmt == "fp16x2":
        hi = x.half().float()
        return hi + (x - hi).half().float()
    if fmt in ("fp16+e4m3", "fp16+e5m2"):
        hi = x.half().float()
        lo = x - hi
        dt = torch.float8_e4m3fn if fmt.endswith("e4m3") else torch.float8_e5m2
        top = 448.0 if fmt.endswith("e4m3") else 57344.0
        m = float(lo.abs().max())
        if m == 0.0:
            return hi
        k = math.floor(math.log2(top / m))           # per-tensor power-of-two scale: the largest |lo| sits just under the format's max
        return hi + (lo * 2.0 ** k).to(dt).float() * 2.0 ** -k
    raise ValueError(fmt)


class Rounding:
    def __init__(self, default: str = "fp16", modes: Optional[Dict[str, str]] = None):
        self.default, self.modes = default, dict(modes or {})

    def __call__(self, site: str, x: Tensor) -> Tensor:
        return _round(x, self.modes.get(site, self.default))


def _mm(R: Rounding, sa: str, a: Tensor, sw: str, w: Tensor, b: Optional[Tensor]) -> Tensor:
    y = R(sa, a) @ R(sw, w).T
    return y if b is None else y + b


def _attn(R: Rounding, pre: str, q: Tensor, k: Tensor, v: Tensor, nheads: int, tail: int = 0) -> Tensor:
    """softmax(q k^T / sqrt(dh)) v on already-projected (and already rounded) q, k, v; P is rounded before the PV product and
    the row sum is taken over the UNROUNDED fp32 numerators, as kernels_attn.h does."""
    B, Lq, d = q.shape
    Lk, dh = k.shape[1], d // nheads
    qh = q.view(B, Lq, nheads, dh).transpose(1, 2)
    kh = k.view(B, Lk, nheads, dh).transpose(1, 2)
    vh = v.view(B, Lk, nheads, dh).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) / math.sqrt(dh)
    if hasattr(R, "logit_peak"):   # tests/tools/trained_like_budget.py: the largest |logit| any attention of the forward sees
        R.logit_peak = max(R.logit_peak, float(s.abs().max()))
    p = torch.exp(s - s.amax(-1, keepdim=True))
    o = (R(pre + ".p", p) @ vh) / p.sum(-1, keepdim=True)
    return o.transpose(1, 2).reshape(B, Lq, d)


class LowPrecDenoiser(O.OracleDenoiser):
    """OracleDenoiser with the GPU's rounding sites.  The hoisted conditioning (K/V of every layer) is computed once per
    (cond_embed, drop) like a2p_prepare_cond does and cached on the object."""

    def __init__(self, sd, data_format, num_layers, num_heads, rounding: Rounding):
        super().__init__(sd, data_format, num_layers, num_heads)
        self.R = rounding
        self._kv = {}

    # -- hoisted conditioning: cond_projection, cond_encoder, norm_cond + rotary, K / V of all layers (a2p_prepare_cond) --
    def _enc_layer(self, p: str, x: Tensor) -> Tensor:
        R, sd, d = self.R, self.sd, self.d
        g = lambda n: sd[p + n]
        xh = O.layer_norm(x, g("norm1.weight"), g("norm1.bias"))
        xr = O.rotary(xh, self.freqs)
        inw, inb = g("self_attn.in_proj_weight"), g("self_attn.in_proj_bias")
        q = R("cond.a", _mm(R, "cond.a", xr, "cond.w", inw[:d], inb[:d]))
        k = R("cond.a", _mm(R, "cond.a", xr, "cond.w", inw[d:2 * d], inb[d:2 * d]))
        v = R("cond.a", _mm(R, "cond.a", xh, "cond.w", inw[2 * d:], inb[2 * d:]))
        Rc = Rounding(self.R.modes.get("cond.a", self.R.default))
        ao = R("cond.a", _attn(Rc, "x", q, k, v, self.H))
        x = x + _mm(R, "cond.a", ao, "cond.w", g("self_attn.out_proj.weight"), g("self_attn.out_proj.bias"))
        xh = O.layer_norm(x, g("norm2.weight"), g("norm2.bias"))
        h = F.gelu(_mm(R, "cond.a", xh, "cond.w", g("linear1.weight"), g("linear1.bias")))
        return x + _mm(R, "cond.a", h, "cond.w", g("linear2.weight"), g("linear2.bias"))

    def _prepare(self, cond_embed: Tensor, drop: float):
        key = (cond_embed.data_ptr(), drop)
        if key in self._kv:
            return self._kv[key]
        R, sd, d = self.R, self.sd, self.d
        if drop == 1.0:
            ct = sd["null_cond_embed"][:, : cond_embed.shape[1], :].expand(cond_embed.shape[0], -1, -1)
            hidden = sd["null_cond_hidden"].expand(cond_embed.shape[0], -1)
        else:
            ct = _mm(R, "cond.a", cond_embed, "cond.w", sd["cond_projection.weight"], sd["cond_projection.bias"])
            for i in range(2 if self.data_format == "face" else 0):
                ct = self._enc_layer(f"cond_encoder.{i}.", ct)
            pooled = ct.mean(dim=-2)                                                   # fp32 skinny path on the GPU
            h = O.layer_norm(pooled, sd["non_attn_cond_projection.0.weight"], sd["non_attn_cond_projection.0.bias"])
            h = F.silu(h @ sd["non_attn_cond_projection.1.weight"].T + sd["non_attn_cond_projection.1.bias"])
            hidden = h @ sd["non_attn_cond_projection.3.weight"].T + sd["non_attn_cond_projection.3.bias"]
        S0 = ct.shape[1]
        # norm_cond over [ct ; time tokens]: LayerNorm is row-local, rotary position = row index
        mem = O.layer_norm(ct, sd["norm_cond.weight"], sd["norm_cond.bias"])
        mem_r = O.rotary(mem, self.freqs)
        ks, vs = [], []
        for l in range(self.L):
            p = f"seqTransDecoder.stack.{l}.multihead_attn."
            inw, inb = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
            ks.append(R("cross.kv", _mm(R, "cond.a", mem_r, "cond.w", inw[d:2 * d], inb[d:2 * d])))
            vs.append(R("cross.kv", _mm(R, "cond.a", mem, "cond.w", inw[2 * d:], inb[2 * d:])))
        self._kv[key] = (ks, vs, hidden, S0)
        return self._kv[key]

    def _prepare_keyframes(self, keyframes: Tensor, mask: Tensor, drop: float):
        """encode_keyframes (model/diffusion.py:315-336) + K / V of multihead_attn2 for every layer (a2p_prepare_cond)."""
        key = ("kf", keyframes.data_ptr(), drop)
        if key in self._kv:
            return self._kv[key]
        R, sd, d = self.R, self.sd, self.d
        if drop == 1.0:
            tok = sd["null_pose_embed"][:, : keyframes.shape[1], :].expand(keyframes.shape[0], -1, -1)
        else:
            pred = keyframes.clone()
            pred[~mask[..., :: self.step].squeeze((1, 2))] = 0.0
            hid = _mm(R, "cond.a", pred, "cond.w", sd["frame_cond_projection.weight"], sd["frame_cond_projection.bias"])
            tok = O.layer_norm(hid, sd["frame_norm_cond.weight"], sd["frame_norm_cond.bias"])
        tok_r = O.rotary(tok, self.freqs)
        ks, vs = [], []
        for l in range(self.L):
            p = f"seqTransDecoder.stack.{l}.multihead_attn2."
            inw, inb = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
            ks.append(R("cross2.kv", _mm(R, "cond.a", tok_r, "cond.w", inw[d:2 * d], inb[d:2 * d])))
            vs.append(R("cross2.kv", _mm(R, "cond.a", tok, "cond.w", inw[2 * d:], inb[2 * d:])))
        self._kv[key] = (ks, vs)
        return self._kv[key]

    def _conv_tail(self, out: Tensor) -> Tensor:
        """pose_conv_tail of the oracle with 16-bit operand staging: every layer's input rows and weights are rounded
        (`tail.a`, `tail.w`); the skip path reads the same rounded rows (a2p_lib_run.h pose_conv_tail)."""
        R, sd = self.R, self.sd
        out = F.pad(R("tail.a", out), pad=[24, 0])
        for i, dil in enumerate((1, 2, 3, 1, 2, 3)):
            y = F.leaky_relu(F.conv1d(out, R("tail.w", sd[f"post_pose_layers.{i}.weight"]), sd[f"post_pose_layers.{i}.bias"],
                                      dilation=dil), negative_slope=0.2)
            out = R("tail.a", (out[:, :, -y.shape[-1]:] + y) / 2.0 if out.shape[1] == y.shape[1] else y)
        return F.conv1d(out, R("tail.w", sd["final_conv.weight"]), sd["final_conv.bias"])

    def forward(self, x, times, cond_embed, keyframes=None, mask=None, cond_drop_prob: float = 0.0) -> Tensor:
        R, sd, d = self.R, self.sd, self.d
        if x.dim() == 4:
            x = x.permute(0, 3, 1, 2).squeeze(-1)
        ks, vs, cond_hidden, S0 = self._prepare(cond_embed, cond_drop_prob)
        pose = self.data_format == "pose"
        if pose:
            k2s, v2s = self._prepare_keyframes(keyframes, mask, cond_drop_prob)
        x = _mm(R, "in.a", x, "in.w", sd["input_projection.weight"], sd["input_projection.bias"])
        emb = O.sinusoidal_pos_emb(times, d, torch.float32)
        t_hidden = O.mish(emb @ sd["time_mlp.1.weight"].T + sd["time_mlp.1.bias"])
        t = t_hidden @ sd["to_time_cond.0.weight"].T + sd["to_time_cond.0.bias"] + cond_hidden
        t_tok = (t_hidden @ sd["to_time_tokens.0.weight"].T + sd["to_time_tokens.0.bias"]).view(-1, 2, d)
        t_tok = O.layer_norm(t_tok, sd["norm_cond.weight"], sd["norm_cond.bias"])
        # rotary of the time tokens at positions S0, S0+1 (they sit behind the audio tokens)
        L_all = S0 + 2
        pad = torch.zeros(t_tok.shape[0], L_all, d)
        pad[:, S0:] = t_tok
        t_tok_r = O.rotary(pad, self.freqs)[:, S0:]
        for l in range(self.L):
            p = f"seqTransDecoder.stack.{l}."
            g = lambda n: sd[p + n]
            # ---- self attention block (PRE chain kernel + attention + MID's out_proj)
            xh = O.layer_norm(x, g("norm1.weight"), g("norm1.bias"))
            xr = O.rotary(xh, self.freqs)
            inw, inb = g("self_attn.in_proj_weight"), g("self_attn.in_proj_bias")
            q = R("self.q", _mm(R, "qkv.a", xr, "qkv.w", inw[:d], inb[:d]))
            k = R("self.k", _mm(R, "qkv.a", xr, "qkv.w", inw[d:2 * d], inb[d:2 * d]))
            v = R("self.v", _mm(R, "qkv.a", xh, "qkv.w", inw[2 * d:], inb[2 * d:]))
            ao = R("self.o", _attn(R, "self", q, k, v, self.H))
            x1 = _mm(R, "self.o", ao, "oself.w", g("self_attn.out_proj.weight"), g("self_attn.out_proj.bias"))
            x = x + O.film_affine(x1, t, g("film1.block.1.weight"), g("film1.block.1.bias"))
            # ---- cross attention block
            xh = O.layer_norm(x, g("norm2.weight"), g("norm2.bias"))
            inw, inb = g("multihead_attn.in_proj_weight"), g("multihead_attn.in_proj_bias")
            q = R("cross.q", _mm(R, "qc.a", O.rotary(xh, self.freqs), "qc.w", inw[:d], inb[:d]))
            kt = R("cross.tail", t_tok_r @ inw[d:2 * d].T + inb[d:2 * d])               # fp32 skinny GEMMs, rounded into the tile
            vt = R("cross.tail", t_tok @ inw[2 * d:].T + inb[2 * d:])
            k = torch.cat((ks[l].expand(x.shape[0], -1, -1), kt), dim=1)
            v = torch.cat((vs[l].expand(x.shape[0], -1, -1), vt), dim=1)
            ao = R("cross.o", _attn(R, "cross", q, k, v, self.H))
            x2 = _mm(R, "cross.o", ao, "ocross.w", g("multihead_attn.out_proj.weight"), g("multihead_attn.out_proj.bias"))
            x = x + O.film_affine(x2, t, g("film2.block.1.weight"), g("film2.block.1.bias"))
            if pose:   # keyframe cross attention (transformer_modules.py:203-208 in the oracle's numbering)
                xh = O.layer_norm(x, g("norm2a.weight"), g("norm2a.bias"))
                inw, inb = g("multihead_attn2.in_proj_weight"), g("multihead_attn2.in_proj_bias")
                q = R("cross2.q", _mm(R, "qc2.a", O.rotary(xh, self.freqs), "qc2.w", inw[:d], inb[:d]))
                ao = R("cross2.o", _attn(R, "cross2", q, k2s[l].expand(x.shape[0], -1, -1), v2s[l].expand(x.shape[0], -1, -1), self.H))
                x2a = _mm(R, "cross2.o", ao, "ocross2.w", g("multihead_attn2.out_proj.weight"), g("multihead_attn2.out_proj.bias"))
                x = x + O.film_affine(x2a, t, g("film2a.block.1.weight"), g("film2a.block.1.bias"))
            # ---- feed-forward
            xh = O.layer_norm(x, g("norm3.weight"), g("norm3.bias"))
            h = F.gelu(_mm(R, "ff1.a", xh, "ff1.w", g("linear1.weight"), g("linear1.bias")))
            x3 = _mm(R, "ff2.a", h, "ff2.w", g("linear2.weight"), g("linear2.bias"))
            x = x + O.film_affine(x3, t, g("film3.block.1.weight"), g("film3.block.1.bias"))
        out = _mm(R, "fin.a", x, "fin.w", sd["final_layer.weight"], sd["final_layer.bias"])
        if pose:
            out = self._conv_tail(out.permute(0, 2, 1)).permute(0, 2, 1)
        return out


ALL_SITES = ["in.a", "in.w", "qkv.a", "qkv.w", "self.q", "self.k", "self.v", "self.p", "self.o", "oself.w", "qc.a", "qc.w",
             "cross.q", "cross.kv", "cross.tail", "cross.p", "cross.o", "ocross.w", "ff1.a", "ff1.w", "ff2.a", "ff2.w",
             "fin.a", "fin.w", "cond.a", "cond.w"]
POSE_SITES = ["qc2.a", "qc2.w", "cross2.q", "cross2.kv", "cross2.p", "cross2.o", "ocross2.w", "tail.a", "tail.w"]
