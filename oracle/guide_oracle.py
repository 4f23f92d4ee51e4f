"""CPU restatement of the guide transformer's sampling path and of the residual-VQ decode.

TEST INFRASTRUCTURE ONLY (same rules as oracle/a2p_oracle.py): imported by tests/ and nothing else.  Pinned against outputs
of the reference itself (tests/golden/make_golden_guide.py -> golden_guide_v1.npz).

Restates, for inference (dropout off, cond_drop_prob in {0, 1}):
  * GuideTransformer.forward            model/guide.py:140-173   (audio features are given: the vq-wav2vec front end,
                                                                   `encode_audio` :111-119, is outside this path)
  * the `pre_audio` conv stack          model/guide.py:84-109    (Conv1d k=3, dilation 1,2,3,1,2,3 per block, no padding,
                                                                   LeakyReLU(0.2), closing 1x1 conv)
  * GuideTransformer.generate           model/guide.py:175-222   (nucleus rule :202-214; the categorical draw takes an
                                                                   injected uniform per step instead of torch's global RNG)
  * FiLMTransformerDecoderLayer with a causal tgt_mask            model/modules/transformer_modules.py:178-267
  * TemporalVertexCodec.decode          model/vqvae.py:508-521, ResidualVectorQuantization.decode :381-392,
                                        TemporalVertexDecoder.forward :452-463
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from .a2p_oracle import film_affine, layer_norm, rotary


def _mha_masked(q_in, k_in, v_in, in_w, in_b, out_w, out_b, nheads, causal: bool):
    """nn.MultiheadAttention as called by _sa_block / _mha_block, with the additive -inf upper-triangular tgt_mask of
    get_tgt_mask (model/guide.py:121-129) when `causal`."""
    d = q_in.shape[-1]
    q = q_in @ in_w[:d].T + in_b[:d]
    k = k_in @ in_w[d:2 * d].T + in_b[d:2 * d]
    v = v_in @ in_w[2 * d:].T + in_b[2 * d:]
    B, Lq, _ = q.shape
    Lk, dh = k.shape[1], d // nheads
    q = q.view(B, Lq, nheads, dh).transpose(1, 2)
    k = k.view(B, Lk, nheads, dh).transpose(1, 2)
    v = v.view(B, Lk, nheads, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    if causal:
        s = s + torch.full((Lq, Lk), float("-inf")).triu(1)
    o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, Lq, d)
    return o @ out_w.T + out_b


def guide_decoder_layer(sd, p, x, memory, t, nheads, freqs):
    g = lambda n: sd[p + n]  # noqa: E731
    xh = layer_norm(x, g("norm1.weight"), g("norm1.bias"))
    qk = rotary(xh, freqs)
    x1 = _mha_masked(qk, qk, xh, g("self_attn.in_proj_weight"), g("self_attn.in_proj_bias"), g("self_attn.out_proj.weight"),
                     g("self_attn.out_proj.bias"), nheads, True)
    x = x + film_affine(x1, t, g("film1.block.1.weight"), g("film1.block.1.bias"))
    xh = layer_norm(x, g("norm2.weight"), g("norm2.bias"))
    x2 = _mha_masked(rotary(xh, freqs), rotary(memory, freqs), memory, g("multihead_attn.in_proj_weight"),
                     g("multihead_attn.in_proj_bias"), g("multihead_attn.out_proj.weight"), g("multihead_attn.out_proj.bias"),
                     nheads, False)
    x = x + film_affine(x2, t, g("film2.block.1.weight"), g("film2.block.1.bias"))
    xh = layer_norm(x, g("norm3.weight"), g("norm3.bias"))
    x3 = F.gelu(xh @ g("linear1.weight").T + g("linear1.bias")) @ g("linear2.weight").T + g("linear2.bias")
    return x + film_affine(x3, t, g("film3.block.1.weight"), g("film3.block.1.bias"))


class OracleGuide:
    def __init__(self, sd: Dict[str, Tensor], tokens: int, num_layers: int, num_heads: int, dilations):
        self.sd, self.tokens, self.L, self.H, self.dil = sd, tokens, num_layers, num_heads, tuple(dilations)
        self.freqs = sd["rotary.freqs"]

    def pre_audio(self, cond_embed: Tensor) -> Tensor:
        """[B, S, C] -> [B, S - 2*sum(dil), C]"""
        x, n = cond_embed.permute(0, 2, 1), 0
        for dl in self.dil:
            x = F.leaky_relu(F.conv1d(x, self.sd[f"pre_audio.{n}.weight"], self.sd[f"pre_audio.{n}.bias"], dilation=dl), 0.2)
            n += 3
        x = F.conv1d(x, self.sd[f"pre_audio.{n}.weight"], self.sd[f"pre_audio.{n}.bias"])
        return x.permute(0, 2, 1)

    def condition(self, cond_embed: Tensor, cond_drop_prob: float = 0.0):
        sd = self.sd
        ct = self.pre_audio(cond_embed) @ sd["cond_projection.weight"].T + sd["cond_projection.bias"]
        if cond_drop_prob == 1.0:
            ct = sd["null_cond_embed"][:, : ct.shape[1]].expand(ct.shape[0], -1, -1)
        h = layer_norm(ct.mean(dim=-2), sd["non_attn_cond_projection.0.weight"], sd["non_attn_cond_projection.0.bias"])
        h = F.silu(h @ sd["non_attn_cond_projection.1.weight"].T + sd["non_attn_cond_projection.1.bias"])
        h = h @ sd["non_attn_cond_projection.3.weight"].T + sd["non_attn_cond_projection.3.bias"]
        if cond_drop_prob == 1.0:
            h = sd["null_cond_hidden"].expand(h.shape[0], -1)
        return layer_norm(ct, sd["norm_cond.weight"], sd["norm_cond.bias"]), h

    def forward(self, tokens: Tensor, cond_embed: Tensor, cond_drop_prob: float = 0.0, cond=None) -> Tensor:
        mem, h = cond if cond is not None else self.condition(cond_embed, cond_drop_prob)
        x = self.sd["token_embedding.weight"][tokens]
        for l in range(self.L):
            x = guide_decoder_layer(self.sd, f"seqTransDecoder.stack.{l}.", x, mem, h, self.H, self.freqs)
        return x @ self.sd["final_layer.weight"].T + self.sd["final_layer.bias"]

    @staticmethod
    def nucleus_probs(logits: Tensor, top_p: float):
        """model/guide.py:201-214: sorted (descending) probabilities with everything after the nucleus zeroed, renormalised."""
        sorted_probs, indices = torch.sort(torch.softmax(logits, dim=-1), dim=-1, descending=True)
        nucleus = torch.cumsum(sorted_probs, dim=-1) < top_p
        nucleus = torch.cat([nucleus.new_ones(nucleus.shape[:-1] + (1,)), nucleus[..., :-1]], dim=-1)
        sorted_probs = sorted_probs.masked_fill(~nucleus, 0.0)
        return sorted_probs / sorted_probs.sum(-1, keepdim=True), indices

    def generate(self, cond_embed: Tensor, sequence_length: int, layers: int, uniforms: Tensor, top_p: float = 0.94) -> Tensor:
        """uniforms [sequence_length * layers, B] in [0, 1): token = first sorted index whose cumulative probability exceeds u
        (the inverse-CDF form of Categorical(sorted_probs).sample())."""
        B = cond_embed.shape[0]
        cond = self.condition(cond_embed)
        toks = torch.full((B, 1), self.tokens, dtype=torch.int64)
        for i in range(sequence_length * layers):
            probs, idx = self.nucleus_probs(self.forward(toks, cond_embed, cond=cond)[:, -1, :], top_p)
            pick = (torch.cumsum(probs, dim=-1) > uniforms[i][:, None]).float().argmax(dim=-1)
            toks = torch.cat([toks, idx.gather(-1, pick[:, None])], dim=-1)
        return toks[:, 1:].contiguous()


def vq_decode(sd: Dict[str, Tensor], q: Tensor, residual_depth: int) -> Tensor:
    """q [B, T, depth] int64 -> [B, T, n_vertices]."""
    enc = sum(sd[f"quantizer.layers.{i}._codebook.embed"][q[..., i]] for i in range(residual_depth))   # [B, T, e]
    x = F.pad(enc.permute(0, 2, 1), [7, 0])                                                            # receptive field 8
    for i, dl in zip((0, 2, 4, 6), (1, 2, 3, 1)):
        x = F.leaky_relu(F.conv1d(x, sd[f"decoder.dec.{i}.weight"], sd[f"decoder.dec.{i}.bias"], dilation=dl), 0.2)
    x = F.conv1d(x, sd["decoder.dec.8.weight"], sd["decoder.dec.8.bias"])
    return x.permute(0, 2, 1)
