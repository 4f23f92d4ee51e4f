// Third part of a2p_lib.hip (same translation unit): the guide transformer context and the residual-VQ decode
// (include/a2p_hip.h "guide" section; reference model/guide.py, model/vqvae.py).
#pragma once
#include "kernels_guide.h"

struct a2p_guide_ctx {
  a2p_guide_config cfg;
  a2p_ctx core;  // fp32 host state for the shared launchers (GEMM dispatch, timers); owns no buffers
  std::map<std::string, int64_t> expect;
  std::map<std::string, Buf> w;
  bool finalized = false, prepared = false;
  int d = 0, H = 0, L = 0, ff = 0, V = 0, C = 0, nconv = 0;
  int pB = 0, pS = 0, pSv = 0;
  std::vector<int> dil;
  std::vector<Buf> conv_w;                         // [tap][Co][Ci]
  Buf rope, wk, bk, wv, bv, film_w, film_b, layers;  // packed per-layer cross K / V projections, FiLM blocks, GuideLayerW[L]
  Buf cbuf[2], ct, mem, memr, pooled, hidden, mish, film, kc, vc, sk, sv;
};

static const float* GW(a2p_guide_ctx* g, const std::string& n) { return g->w.at(n).f(); }

extern "C" int a2p_guide_create(const a2p_guide_config* cfg, a2p_guide_ctx** out) {
  ARG(cfg && out, "null argument");
  ARG(cfg->dim % 64 == 0 && cfg->dim <= 512 && cfg->num_heads > 0 && cfg->dim % cfg->num_heads == 0 &&
          (cfg->dim / cfg->num_heads) % 4 == 0,
      "dim must be a multiple of 64 (<= 512) and of 4 * num_heads (got %d / %d)", cfg->dim, cfg->num_heads);
  ARG(cfg->ff_size % 64 == 0 && cfg->cond_feature_dim % 64 == 0 && cfg->tokens >= 2 && cfg->tokens <= 4096, "bad ff / cond / tokens");
  ARG(cfg->num_layers >= 1 && cfg->num_audio_layers >= 0 && cfg->max_batch >= 1 && cfg->max_positions >= 1 && cfg->emb_len >= 16,
      "bad capacity");
  a2p_guide_ctx* g = new a2p_guide_ctx();
  g->cfg = *cfg;
  g->d = cfg->dim; g->H = cfg->num_heads; g->L = cfg->num_layers; g->ff = cfg->ff_size; g->V = cfg->tokens; g->C = cfg->cond_feature_dim;
  g->core.bf16 = false; g->core.esz = 4; g->core.d = g->d; g->core.use_arena = false;
  for (int a = 0; a < cfg->num_audio_layers; ++a)
    for (int dl : {1, 2, 3, 1, 2, 3}) g->dil.push_back(dl);  // model/guide.py:84-109
  g->nconv = (int)g->dil.size() + 1;
  const int64_t d = g->d, ff = g->ff, C = g->C;
  auto& e = g->expect;
  e["token_embedding.weight"] = (int64_t)(g->V + 1) * d;
  e["null_cond_embed"] = (int64_t)cfg->emb_len * d;
  e["null_cond_hidden"] = d;
  e["norm_cond.weight"] = d; e["norm_cond.bias"] = d;
  e["cond_projection.weight"] = d * C; e["cond_projection.bias"] = d;
  e["non_attn_cond_projection.0.weight"] = d; e["non_attn_cond_projection.0.bias"] = d;
  e["non_attn_cond_projection.1.weight"] = d * d; e["non_attn_cond_projection.1.bias"] = d;
  e["non_attn_cond_projection.3.weight"] = d * d; e["non_attn_cond_projection.3.bias"] = d;
  for (int i = 0; i < g->nconv; ++i) {
    const std::string p = "pre_audio." + std::to_string(3 * i) + ".";
    e[p + "weight"] = C * C * (i + 1 < g->nconv ? 3 : 1);
    e[p + "bias"] = C;
  }
  for (int l = 0; l < g->L; ++l) {
    const std::string p = "seqTransDecoder.stack." + std::to_string(l) + ".";
    for (const char* a : {"self_attn", "multihead_attn"}) {
      e[p + a + ".in_proj_weight"] = 3 * d * d; e[p + a + ".in_proj_bias"] = 3 * d;
      e[p + a + ".out_proj.weight"] = d * d; e[p + a + ".out_proj.bias"] = d;
    }
    e[p + "linear1.weight"] = ff * d; e[p + "linear1.bias"] = ff;
    e[p + "linear2.weight"] = d * ff; e[p + "linear2.bias"] = d;
    for (const char* n : {"norm1", "norm2", "norm3"}) { e[p + n + ".weight"] = d; e[p + n + ".bias"] = d; }
    for (const char* f : {"film1", "film2", "film3"}) { e[p + f + ".block.1.weight"] = 2 * d * d; e[p + f + ".block.1.bias"] = 2 * d; }
  }
  e["final_layer.weight"] = (int64_t)g->V * d; e["final_layer.bias"] = g->V;
  *out = g;
  return 0;
}

extern "C" int a2p_guide_destroy(a2p_guide_ctx* g) {
  if (!g) return 0;
  (void)hipDeviceSynchronize();
  for (auto& kv : g->w) buf_free(kv.second);
  for (auto& b : g->conv_w) buf_free(b);
  Buf* all[] = {&g->rope, &g->wk, &g->bk, &g->wv, &g->bv, &g->film_w, &g->film_b, &g->layers, &g->cbuf[0], &g->cbuf[1], &g->ct, &g->mem,
                &g->memr, &g->pooled, &g->hidden, &g->mish, &g->film, &g->kc, &g->vc, &g->sk, &g->sv};
  for (Buf* b : all) buf_free(*b);
  delete g;
  return 0;
}

extern "C" int a2p_guide_set_weight(a2p_guide_ctx* g, const char* name, const float* dev_ptr, int64_t numel, void* stream) {
  ARG(g && name && dev_ptr, "null argument");
  const std::string n(name);
  if (n.size() >= 12 && n.compare(n.size() - 12, 12, "rotary.freqs") == 0) return 0;  // recomputed from dim (rotary_embedding_torch.py:99-101)
  if (n.rfind("audio_model.", 0) == 0) return 0;                                      // the vq-wav2vec front end is outside this path
  auto it = g->expect.find(n);
  if (it == g->expect.end()) {
    set_err("unexpected guide parameter '%s'", name);
    return A2P_ERR_NOWEIGHT;
  }
  if (it->second != numel) {
    set_err("guide parameter '%s': expected %lld elements, got %lld", name, (long long)it->second, (long long)numel);
    return A2P_ERR_NOWEIGHT;
  }
  Buf& b = g->w[n];
  if (!b.p) CHK(buf_alloc_tmp(b, (size_t)numel * 4));
  HIPCHK(hipMemcpyAsync(b.p, dev_ptr, (size_t)numel * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  g->finalized = false;
  return 0;
}

extern "C" int a2p_guide_finalize(a2p_guide_ctx* g, void* stream) {
  ARG(g, "null argument");
  hipStream_t s = (hipStream_t)stream;
  for (auto& kv : g->expect)
    if (!g->w.count(kv.first)) {
      set_err("guide parameter '%s' was never set", kv.first.c_str());
      return A2P_ERR_NOWEIGHT;
    }
  const int d = g->d, L = g->L, C = g->C;
  const int npos = std::max(g->cfg.emb_len, g->cfg.max_positions) + 8;
  {  // rotary table from freqs = 1 / 10000^(2i/d)
    std::vector<float> fr(d / 2);
    for (int i = 0; i < d / 2; ++i) fr[i] = 1.0f / powf(10000.0f, (float)(2 * i) / (float)d);
    Buf tmp;
    CHK(buf_alloc_tmp(tmp, fr.size() * 4));
    HIPCHK(hipMemcpyAsync(tmp.p, fr.data(), fr.size() * 4, hipMemcpyHostToDevice, s));
    CHK(buf_alloc_tmp(g->rope, (size_t)npos * (d / 2) * 8));
    rope_table_kernel<<<(npos * (d / 2) + 255) / 256, 256, 0, s>>>(tmp.f(), (float2*)g->rope.p, npos, d / 2);
    HIPCHK(hipStreamSynchronize(s));
    buf_free(tmp);
  }
  g->conv_w.resize(g->nconv);
  for (int i = 0; i < g->nconv; ++i) {
    const int taps = i + 1 < g->nconv ? 3 : 1;
    CHK(buf_alloc_tmp(g->conv_w[i], (size_t)C * C * taps * 4));
    const int64_t n = (int64_t)C * C * taps;
    guide_conv_repack_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>(GW(g, "pre_audio." + std::to_string(3 * i) + ".weight"),
                                                                    g->conv_w[i].f(), C, C, taps);
  }
  CHK(buf_alloc_tmp(g->wk, (size_t)L * d * d * 4)); CHK(buf_alloc_tmp(g->wv, (size_t)L * d * d * 4));
  CHK(buf_alloc_tmp(g->bk, (size_t)L * d * 4)); CHK(buf_alloc_tmp(g->bv, (size_t)L * d * 4));
  CHK(buf_alloc_tmp(g->film_w, (size_t)L * 3 * 2 * d * d * 4)); CHK(buf_alloc_tmp(g->film_b, (size_t)L * 3 * 2 * d * 4));
  std::vector<GuideLayerW> lw(L);
  for (int l = 0; l < L; ++l) {
    const std::string p = "seqTransDecoder.stack." + std::to_string(l) + ".";
    const float* cin = GW(g, p + "multihead_attn.in_proj_weight");
    const float* cib = GW(g, p + "multihead_attn.in_proj_bias");
    auto cp = [&](float* dst, const float* src, size_t n) { return hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToDevice, s); };
    HIPCHK(cp(g->wk.f() + (size_t)l * d * d, cin + (size_t)d * d, (size_t)d * d));
    HIPCHK(cp(g->wv.f() + (size_t)l * d * d, cin + (size_t)2 * d * d, (size_t)d * d));
    HIPCHK(cp(g->bk.f() + (size_t)l * d, cib + d, d));
    HIPCHK(cp(g->bv.f() + (size_t)l * d, cib + 2 * d, d));
    const char* films[3] = {"film1", "film2", "film3"};
    for (int f = 0; f < 3; ++f) {
      HIPCHK(cp(g->film_w.f() + ((size_t)l * 3 + f) * 2 * d * d, GW(g, p + films[f] + ".block.1.weight"), (size_t)2 * d * d));
      HIPCHK(cp(g->film_b.f() + ((size_t)l * 3 + f) * 2 * d, GW(g, p + films[f] + ".block.1.bias"), (size_t)2 * d));
    }
    GuideLayerW& w = lw[l];
    w.ln1_g = GW(g, p + "norm1.weight"); w.ln1_b = GW(g, p + "norm1.bias");
    w.sa_in_w = GW(g, p + "self_attn.in_proj_weight"); w.sa_in_b = GW(g, p + "self_attn.in_proj_bias");
    w.sa_out_w = GW(g, p + "self_attn.out_proj.weight"); w.sa_out_b = GW(g, p + "self_attn.out_proj.bias");
    w.ln2_g = GW(g, p + "norm2.weight"); w.ln2_b = GW(g, p + "norm2.bias");
    w.ca_q_w = cin; w.ca_q_b = cib;
    w.ca_out_w = GW(g, p + "multihead_attn.out_proj.weight"); w.ca_out_b = GW(g, p + "multihead_attn.out_proj.bias");
    w.ln3_g = GW(g, p + "norm3.weight"); w.ln3_b = GW(g, p + "norm3.bias");
    w.w1 = GW(g, p + "linear1.weight"); w.b1 = GW(g, p + "linear1.bias");
    w.w2 = GW(g, p + "linear2.weight"); w.b2 = GW(g, p + "linear2.bias");
  }
  CHK(buf_alloc_tmp(g->layers, lw.size() * sizeof(GuideLayerW)));
  HIPCHK(hipMemcpyAsync(g->layers.p, lw.data(), lw.size() * sizeof(GuideLayerW), hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  const int B = g->cfg.max_batch, T = g->cfg.max_positions;
  CHK(buf_alloc_tmp(g->sk, (size_t)B * L * T * d * 4)); CHK(buf_alloc_tmp(g->sv, (size_t)B * L * T * d * 4));
  g->finalized = true;
  g->prepared = false;
  return 0;
}

// Everything of GuideTransformer.forward that does not depend on the tokens (model/guide.py:150-169)
extern "C" int a2p_guide_prepare(a2p_guide_ctx* g, const float* cond_embed, int32_t batch, int32_t n_tokens, int32_t cond_drop,
                                 void* stream) {
  ARG(g && cond_embed, "null argument");
  ARG(g->finalized, "a2p_guide_finalize has not been called");
  int shrink = 0;
  for (int dl : g->dil) shrink += 2 * dl;
  const int B = batch, S = n_tokens, Sv = S - shrink, d = g->d, L = g->L, C = g->C;
  ARG(B >= 1 && B <= g->cfg.max_batch, "batch %d outside [1, %d]", B, g->cfg.max_batch);
  ARG(Sv >= 1 && Sv <= g->cfg.emb_len, "%d audio tokens leave %d after the conv stack (need 1..emb_len=%d)", S, Sv, g->cfg.emb_len);
  hipStream_t s = (hipStream_t)stream;
  const int64_t R = (int64_t)B * S;
  if (g->pB != B || g->pS != S) {
    const size_t rows = (size_t)R + 8;  // the last rows' taps read up to 6 rows past the end
    CHK(buf_alloc_tmp(g->cbuf[0], rows * C * 4)); CHK(buf_alloc_tmp(g->cbuf[1], rows * C * 4));
    CHK(buf_alloc_tmp(g->ct, (size_t)R * d * 4)); CHK(buf_alloc_tmp(g->mem, (size_t)R * d * 4)); CHK(buf_alloc_tmp(g->memr, (size_t)R * d * 4));
    CHK(buf_alloc_tmp(g->pooled, (size_t)B * d * 4)); CHK(buf_alloc_tmp(g->hidden, (size_t)B * d * 4)); CHK(buf_alloc_tmp(g->mish, (size_t)B * d * 4));
    CHK(buf_alloc_tmp(g->film, (size_t)B * L * 3 * 2 * d * 4));
    CHK(buf_alloc_tmp(g->kc, (size_t)R * L * d * 4)); CHK(buf_alloc_tmp(g->vc, (size_t)R * L * d * 4));
    g->pB = B; g->pS = S;
  }
  g->pSv = Sv;
  g->prepared = false;
  const float* tokens_src = nullptr;
  if (!cond_drop) {
    // pre_audio: valid (unpadded) dilated convs over each sequence's rows; rows past a sequence's shrinking valid length hold
    // finite garbage that no valid row ever reads (a valid output row t only reads input rows t .. t + 2*dilation < valid length)
    // (the input is copied into the padded ping-pong buffer first: the taps of a sequence's last rows read past row R)
    HIPCHK(hipMemcpyAsync(g->cbuf[1].p, cond_embed, (size_t)R * C * 4, hipMemcpyDeviceToDevice, s));
    const float* src = g->cbuf[1].f();
    int cur = 0;
    for (int i = 0; i < g->nconv; ++i) {
      const bool last = i + 1 == g->nconv;
      const std::string pn = "pre_audio." + std::to_string(3 * i) + ".";
      float* dst = g->cbuf[cur].f();
      GemmP p = gemm_base(src, C, g->conv_w[i].p, C, GW(g, pn + "bias"), dst, C, (int)R, C, C);
      p.ntaps = last ? 1 : 3;
      p.a_tap_stride = last ? 0 : (int64_t)g->dil[i] * C;
      p.w_tap_stride = (int64_t)C * C;
      p.epi = EPI_CONV;
      p.act = last ? ACT_NONE : ACT_LRELU;
      if (last) p.epi = EPI_STORE;
      CHK(launch_gemm(&g->core, p, s));
      src = dst;
      cur ^= 1;
    }
    GemmP pj = gemm_base(src, C, GW(g, "cond_projection.weight"), C, GW(g, "cond_projection.bias"), g->ct.p, d, (int)R, d, C);
    CHK(launch_gemm(&g->core, pj, s));
    tokens_src = g->ct.f();
    guide_mean_kernel<<<dim3((d + 255) / 256, B), 256, 0, s>>>(g->ct.f(), g->pooled.f(), S, Sv, d);
  } else {
    HIPCHK(hipMemsetAsync(g->pooled.p, 0, (size_t)B * d * 4, s));
  }
  GuideHiddenP hp;
  hp.pooled = g->pooled.f();
  hp.ln_g = GW(g, "non_attn_cond_projection.0.weight"); hp.ln_b = GW(g, "non_attn_cond_projection.0.bias");
  hp.w1 = GW(g, "non_attn_cond_projection.1.weight"); hp.b1 = GW(g, "non_attn_cond_projection.1.bias");
  hp.w3 = GW(g, "non_attn_cond_projection.3.weight"); hp.b3 = GW(g, "non_attn_cond_projection.3.bias");
  hp.null_hidden = GW(g, "null_cond_hidden"); hp.d = d; hp.drop = cond_drop ? 1 : 0;
  hp.hidden = g->hidden.f(); hp.mish_hidden = g->mish.f();
  guide_hidden_kernel<<<B, 256, 0, s>>>(hp);
  // every DenseFiLM of every layer in one skinny GEMM: [B, d] x [L*3*2d, d]^T
  CHK(launch_skinny(g->mish.f(), d, g->film_w.f(), d, g->film_b.f(), g->film.f(), (int64_t)L * 3 * 2 * d, B, L * 3 * 2 * d, d, ACT_NONE, s));
  guide_memory_kernel<<<(int)((R + 3) / 4), 256, 0, s>>>(tokens_src, cond_drop ? GW(g, "null_cond_embed") : nullptr, GW(g, "norm_cond.weight"),
                                                        GW(g, "norm_cond.bias"), (const float2*)g->rope.p, g->mem.f(), g->memr.f(), nullptr,
                                                        S, Sv, d, (int)R);
  GemmP pk = gemm_base(g->memr.p, d, g->wk.p, d, g->bk.f(), g->kc.p, (int64_t)L * d, (int)R, L * d, d);
  CHK(launch_gemm(&g->core, pk, s));
  GemmP pv = gemm_base(g->mem.p, d, g->wv.p, d, g->bv.f(), g->vc.p, (int64_t)L * d, (int)R, L * d, d);
  CHK(launch_gemm(&g->core, pv, s));
  HIPCHK(hipGetLastError());
  g->prepared = true;
  return 0;
}

static int guide_run(a2p_guide_ctx* g, GuideArP& p, int B, hipStream_t s) {
  ARG(g->prepared, "a2p_guide_prepare has not been called");
  ARG(B == g->pB, "batch %d differs from the prepared batch %d", B, g->pB);
  ARG(p.n_pos >= 1 && p.n_pos <= g->cfg.max_positions, "%d positions outside [1, max_positions=%d]", p.n_pos, g->cfg.max_positions);
  const int d = g->d;
  p.d = d; p.H = g->H; p.L = g->L; p.ff = g->ff; p.V = g->V;
  p.Vp = 1;
  while (p.Vp < p.V) p.Vp <<= 1;
  p.Sv = g->pSv; p.S = g->pS; p.maxT = g->cfg.max_positions;
  p.sc_ld = std::max(p.Sv, p.maxT);
  p.start_token = g->V;
  p.layers = reinterpret_cast<const GuideLayerW*>(g->layers.p);
  p.tok_emb = GW(g, "token_embedding.weight"); p.fin_w = GW(g, "final_layer.weight"); p.fin_b = GW(g, "final_layer.bias");
  p.cs = (const float2*)g->rope.p; p.film = g->film.f(); p.kc = g->kc.f(); p.vc = g->vc.f(); p.sk = g->sk.f(); p.sv = g->sv.f();
  const size_t lds = ((size_t)8 * d + g->ff + std::max(4 * d, 1024) + 64 + (size_t)g->H * p.sc_ld + 2 * (size_t)p.Vp) * 4;
  ARG(lds <= 160 * 1024, "guide_ar_kernel needs %zu bytes of LDS", lds);
  static bool attr_set = false;
  if (!attr_set) {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(guide_ar_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  guide_ar_kernel<<<B, 256, lds, s>>>(p);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int a2p_guide_forward(a2p_guide_ctx* g, const int64_t* tokens, int32_t batch, int32_t len, float* logits, void* stream) {
  ARG(g && tokens && logits, "null argument");
  GuideArP p;
  memset(&p, 0, sizeof(p));
  p.mode = 0; p.n_pos = len; p.tokens_in = tokens; p.logits_out = logits;
  return guide_run(g, p, batch, (hipStream_t)stream);
}

extern "C" int a2p_guide_generate(a2p_guide_ctx* g, int32_t batch, int32_t n_steps, float top_p, const float* uniforms, int64_t* tokens_out,
                                  float* sorted_probs_out, void* stream) {
  ARG(g && uniforms && tokens_out, "null argument");
  GuideArP p;
  memset(&p, 0, sizeof(p));
  p.mode = 1; p.n_pos = n_steps; p.top_p = top_p; p.uniforms = uniforms; p.tokens_out = tokens_out; p.probs_out = sorted_probs_out;
  return guide_run(g, p, batch, (hipStream_t)stream);
}

extern "C" int a2p_guide_debug_read(a2p_guide_ctx* g, const char* name, void* host, int64_t bytes) {
  ARG(g && name && host, "null argument");
  const std::string n(name);
  const int last = (g->nconv - 1) & 1;  // the buffer the closing 1x1 conv wrote
  Buf* b = n == "pre_audio" ? &g->cbuf[last] : n == "ct" ? &g->ct : n == "mem" ? &g->mem : n == "memr" ? &g->memr : n == "hidden" ? &g->hidden
           : n == "film" ? &g->film : n == "kc" ? &g->kc : n == "vc" ? &g->vc : nullptr;
  ARG(b && b->p, "unknown or unallocated guide buffer '%s'", name);
  ARG((size_t)bytes <= b->bytes, "guide buffer '%s' holds %zu bytes", name, b->bytes);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(host, b->p, (size_t)bytes, hipMemcpyDeviceToHost));
  return 0;
}

// TemporalVertexCodec.decode (model/vqvae.py:508-521); all pointers are device pointers, the arrays of pointers live on the host
extern "C" int a2p_vq_decode(const int64_t* q, int32_t batch, int32_t T, int32_t depth, int32_t categories, int32_t latent, int32_t vertices,
                             const float* const* codebooks, const float* const* conv_w, const float* const* conv_b, float* out, void* stream) {
  ARG(q && codebooks && conv_w && conv_b && out, "null argument");
  ARG(depth >= 1 && depth <= 8 && batch >= 1 && T >= 1 && latent >= 1 && vertices >= 1 && categories >= 1, "bad VQ shape");
  VqDecodeP p;
  memset(&p, 0, sizeof(p));
  p.q = q; p.T = T; p.depth = depth; p.e = latent; p.nv = vertices; p.out = out;
  for (int i = 0; i < depth; ++i) p.codebook[i] = codebooks[i];
  for (int i = 0; i < 5; ++i) { p.cw[i] = conv_w[i]; p.cb[i] = conv_b[i]; }
  const size_t lds = (size_t)2 * (T + 7) * latent * 4;
  ARG(lds <= 64 * 1024, "VQ decode of %d frames x %d latent needs %zu bytes of LDS", T, latent, lds);
  vq_decode_kernel<<<batch, 256, lds, (hipStream_t)stream>>>(p);
  HIPCHK(hipGetLastError());
  return 0;
}
