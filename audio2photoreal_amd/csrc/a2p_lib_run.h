// Second half of a2p_lib.hip (same translation unit): hoisted conditioning, the per-step time path,
// the decoder stack, the fused sampler step and the unit / measurement entry points.
#pragma once

// ------------------------------------------------------------------------------------------------
// one norm-first transformer block pieces
// ------------------------------------------------------------------------------------------------
struct CrossKV {
  const void* K = nullptr;
  const void* VT = nullptr;
  int64_t k_slot_stride = 0, ldk = 0, vt_slot_stride = 0, ldvt = 0;
  const int* slots = nullptr;
  int slot_rule = 0, slot_b = 0;   // the table's contents as a formula (AttnP::slot_rule): saves the kernels a dependent load
  int S_main = 0;
  const float* ktail = nullptr;
  const float* vtail = nullptr;
  int64_t tail_sample_stride = 0, tail_row_stride = 0;
  int S_tail = 0, tail_mod = 1;
};

struct FilmRef {
  const float* base = nullptr;  // scale of (layer, film 0) for sequence 0; NULL = plain residual
  int64_t seq_stride = 0;
};

static int film_gemm(a2p_ctx* c, const void* A, int64_t lda, const std::string& wname, const float* bias, int K, const FilmRef& fr,
                     int film_idx, int M, int rows_per_seq, hipStream_t s) {
  const int d = c->d;
  GemmP p = gemm_base(A, lda, c->wt.at(wname).p, rup(K, 64), bias, nullptr, 0, M, d, K);
  p.epi = EPI_FILM_RES;
  p.resid = c->x.f();
  p.ldx = d;
  p.rows_per_seq = rows_per_seq;
  if (fr.base) {
    p.film = fr.base + (int64_t)film_idx * 2 * d;
    p.film_seq_stride = fr.seq_stride;
    p.film_shift_off = d;
  }
  return launch_gemm(c, p, s);
}

// self attention block on the residual stream c->x: x += FiLM(out_proj(MHA(rot(LN x), rot(LN x), LN x)))
static int self_attn_block(a2p_ctx* c, const std::string& p, const std::string& norm, int N, int T, const FilmRef& fr, int film_idx,
                           hipStream_t s) {
  const int d = c->d, M = N * T;
  const int Tld = rup(T, 64);
  CHK(launch_ln_rope(c, false, c->x.f(), d, W32(c, norm + ".weight"), W32(c, norm + ".bias"), c->xn.p, c->xr.p, d, M, T, 0, s));
  const Buf& inw = c->wt.at(p + ".in_proj_weight");
  const float* inb = W32(c, p + ".in_proj_bias");
  GemmP pq = gemm_base(c->xr.p, d, inw.p, d, inb, c->qk.p, 2 * d, M, 2 * d, d);
  CHK(launch_gemm(c, pq, s));
  GemmP pv = gemm_base(c->xn.p, d, c->offT(inw, (int64_t)2 * d * d), d, inb + 2 * d, c->vt.p, Tld, M, d, d);
  pv.epi = EPI_STORE_T;
  pv.rows_per_seq = T;
  pv.t_seq_stride = (int64_t)d * Tld;
  CHK(launch_gemm(c, pv, s));
  AttnP a;
  memset(&a, 0, sizeof(a));
  a.Q = c->qk.p; a.q_seq_stride = (int64_t)T * 2 * d; a.ldq = 2 * d;
  a.K = c->offT(c->qk, d); a.k_slot_stride = (int64_t)T * 2 * d; a.ldk = 2 * d;
  a.VT = c->vt.p; a.vt_slot_stride = (int64_t)d * Tld; a.ldvt = Tld;
  a.O = c->ao.p; a.o_seq_stride = (int64_t)T * d; a.ldo = d;
  a.tail_mod = 1; a.Tq = T; a.S_main = T; a.S_tail = 0;
  a.scale_log2e = 1.4426950408889634f / sqrtf((float)c->DH);
  CHK(launch_attn(c, a, N, A2P_KERNEL_ATTN_SELF, s));
  return film_gemm(c, c->ao.p, d, p + ".out_proj.weight", W32(c, p + ".out_proj.bias"), d, fr, film_idx, M, T, s);
}

static int cross_attn_block(a2p_ctx* c, const std::string& p, const std::string& norm, int N, int T, const CrossKV& kv,
                            const FilmRef& fr, int film_idx, hipStream_t s) {
  const int d = c->d, M = N * T;
  CHK(launch_ln_rope(c, false, c->x.f(), d, W32(c, norm + ".weight"), W32(c, norm + ".bias"), nullptr, c->xr.p, d, M, T, 0, s));
  GemmP pq = gemm_base(c->xr.p, d, c->wt.at(p + ".in_proj_weight").p, d, W32(c, p + ".in_proj_bias"), c->qk.p, d, M, d, d);
  CHK(launch_gemm(c, pq, s));
  AttnP a;
  memset(&a, 0, sizeof(a));
  a.Q = c->qk.p; a.q_seq_stride = (int64_t)T * d; a.ldq = d;
  a.K = kv.K; a.k_slot_stride = kv.k_slot_stride; a.ldk = kv.ldk;
  a.VT = kv.VT; a.vt_slot_stride = kv.vt_slot_stride; a.ldvt = kv.ldvt;
  a.O = c->ao.p; a.o_seq_stride = (int64_t)T * d; a.ldo = d;
  a.ktail = kv.ktail; a.vtail = kv.vtail; a.tail_sample_stride = kv.tail_sample_stride; a.tail_row_stride = kv.tail_row_stride;
  a.kv_slot = kv.slots; a.tail_mod = kv.tail_mod; a.Tq = T; a.S_main = kv.S_main; a.S_tail = kv.S_tail;
  a.slot_rule = kv.slot_rule; a.slot_b = kv.slot_b;
  a.kv_stream = kv.slots && !c->opt.kv_cached ? 1 : 0;
  a.scale_log2e = 1.4426950408889634f / sqrtf((float)c->DH);
  CHK(launch_attn(c, a, N, A2P_KERNEL_ATTN_CROSS, s));
  return film_gemm(c, c->ao.p, d, p + ".out_proj.weight", W32(c, p + ".out_proj.bias"), d, fr, film_idx, M, T, s);
}

static int ffn_block(a2p_ctx* c, const std::string& p, const std::string& norm, int M, int rows_per_seq, const FilmRef& fr,
                     int film_idx, hipStream_t s) {
  const int d = c->d;
  CHK(launch_ln_rope(c, false, c->x.f(), d, W32(c, norm + ".weight"), W32(c, norm + ".bias"), c->xn.p, nullptr, d, M, rows_per_seq, 0, s));
  GemmP p1 = gemm_base(c->xn.p, d, c->wt.at(p + "linear1.weight").p, d, W32(c, p + "linear1.bias"), c->hff.p, c->ff, M, c->ff, d);
  p1.act = ACT_GELU;
  CHK(launch_gemm(c, p1, s));
  return film_gemm(c, c->hff.p, c->ff, p + "linear2.weight", W32(c, p + "linear2.bias"), c->ff, fr, film_idx, M, rows_per_seq, s);
}

// ------------------------------------------------------------------------------------------------
// bf16 throughput mode: the decoder layer as row-panel chain kernels (kernels_chain.h) around the attentions
// ------------------------------------------------------------------------------------------------
static bool chain_supported(const a2p_ctx* c) {
  return c->bf16 && (c->d == 512 || c->d == 256) && c->ff == 1024 && !c->ch_stream.empty() && !c->opt.no_chain;
}

enum { CH_PRE = 0, CH_MID = 1, CH_MID2 = 2, CH_POST = 3, CH_MIDPOST = 4, CH_KINDS = 5 };   // CH_MIDPOST: MID2 | keyframe attention | POST as one kernel (body model)
static int ch_index(int layer, int kind) { return layer * CH_KINDS + kind; }

// stages of one GEMM: 128-row tiles of W[nrows, ldw] (tile-major), K/64 k-steps each
// omap: the GEMM's tiles are stored straight to HBM by chain_body::gemm_store (Q|K, V, Q projections): paired column map for
// the 8-wave slices (chain_pack_kernel)
// group > 0: the tiles are consumed in groups of `group` tiles, k-major inside a group (chain_body::gemm_group)
static void pk_gemm(std::vector<ChainPackDesc>& v, const void* W, int ldw, int nrows, int K, int omap = 0, int group = 0) {
  const int nt = (nrows + 127) / 128;
  if (group <= 0) {
    for (int t = 0; t < nt; ++t)
      for (int ks = 0; ks < K / 64; ++ks) v.push_back({reinterpret_cast<const h16_t*>(W), ldw, t * 128, ks * 64, nrows, omap});
    return;
  }
  for (int t0 = 0; t0 < nt; t0 += group)
    for (int ks = 0; ks < K / 64; ++ks)
      for (int t = t0; t < t0 + group && t < nt; ++t) v.push_back({reinterpret_cast<const h16_t*>(W), ldw, t * 128, ks * 64, nrows, omap});
}

static int chain_pack(a2p_ctx* c, int idx, const std::vector<ChainPackDesc>& descs, const std::vector<std::pair<const float*, int>>& aux,
                      hipStream_t s) {
  const size_t pad = CHAIN_STREAM_PAD;  // the DMA runs up to NS-1 (<= 5) stages past the end
  Buf dd;
  CHK(buf_alloc_tmp(dd, descs.size() * sizeof(ChainPackDesc)));
  HIPCHK(hipMemcpyAsync(dd.p, descs.data(), descs.size() * sizeof(ChainPackDesc), hipMemcpyHostToDevice, s));
  for (int w8 = 0; w8 < 2; ++w8) {  // both slice layouts: 4 waves x 32 out-cols and 8 waves x 16 (chain_pick_nw chooses per box)
    Buf& st = c->ch_stream[(size_t)w8 * c->L * CH_KINDS + idx];
    CHK(buf_alloc(st, (descs.size() + pad) * CHAIN_STAGE_ELEMS * 2));
    chain_pack_kernel<<<(int)descs.size(), 256, 0, s>>>(reinterpret_cast<const ChainPackDesc*>(dd.p), reinterpret_cast<h16_t*>(st.p), w8 ? 8 : 4);
  }
  HIPCHK(hipGetLastError());
  Buf& ax = c->ch_aux[idx];
  CHK(buf_alloc(ax, 2560 * 4 + 1024));
  size_t off = 0;
  for (auto& a : aux) {
    HIPCHK(hipMemcpyAsync(ax.f() + off, a.first, (size_t)a.second * 4, hipMemcpyDeviceToDevice, s));
    off += a.second;
  }
  HIPCHK(hipStreamSynchronize(s));  // descs is host memory of the caller
  buf_free(dd);
  return 0;
}

// kernels_chain4.h: HALF stages (128 output columns x 32 k) in the order gemm<NTG, NKC> consumes them: k-chunk-major inside a group
// of `group` tiles
static void pk4_gemm(std::vector<ChainPackDesc>& v, const void* W, int ldw, int nrows, int k0, int K, int omap, int group, int tile0 = 0, int ntiles = -1) {
  const int nt = ntiles < 0 ? (nrows + 127) / 128 : ntiles;
  for (int t0 = 0; t0 < nt; t0 += group)
    for (int kc = 0; kc < K / 32; ++kc)
      for (int t = t0; t < t0 + group && t < nt; ++t)
        v.push_back({reinterpret_cast<const h16_t*>(W), ldw, (tile0 + t) * 128, k0 + kc * 32, nrows, omap});
}
// descs_e: the order in which waves 0-3 consume the stream where it differs from descs (waves 4-7): the pipelined feed-forward block of the POST kernels
static int chain4_pack(a2p_ctx* c, Buf& st, const std::vector<ChainPackDesc>& descs, hipStream_t s, const std::vector<ChainPackDesc>* descs_e = nullptr) {
  const size_t pad = 2 * CHAIN_STREAM_PAD;   // the register ring runs CHAIN4_PF half stages past the end
  const size_t n = descs.size();
  ARG(!descs_e || descs_e->size() == n, "chain4_pack: the two consumption orders differ in length");
  Buf dd;
  CHK(buf_alloc_tmp(dd, 2 * n * sizeof(ChainPackDesc)));
  HIPCHK(hipMemcpyAsync(dd.p, descs.data(), n * sizeof(ChainPackDesc), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(static_cast<char*>(dd.p) + n * sizeof(ChainPackDesc), (descs_e ? descs_e : &descs)->data(), n * sizeof(ChainPackDesc), hipMemcpyHostToDevice, s));
  CHK(buf_alloc(st, (n + pad) * CHAIN4_HS_ELEMS * 2));
  chain4_pack_kernel<<<(int)n, 256, 0, s>>>(reinterpret_cast<const ChainPackDesc*>(dd.p), reinterpret_cast<h16_t*>(st.p), (int)n);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(s));
  buf_free(dd);
  return 0;
}

// pre-pack every chain's weight stream in consumption order (called from a2p_finalize_weights)
static int chain_build_streams(a2p_ctx* c, hipStream_t s) {
  const int d = c->d, ff = c->ff, L = c->L;
  if (c->ch_stream.size() != (size_t)L * 2 * CH_KINDS) {  // first build; later builds (weight updates) refill the same buffers
    c->ch_stream.assign((size_t)L * 2 * CH_KINDS, Buf());
    c->ch_aux.assign((size_t)L * CH_KINDS, Buf());
    c->ch_stream4.assign((size_t)L * CH_KINDS, Buf());
    c->ch_stream4w.assign((size_t)L, Buf());
  }
  const bool v4 = d == 512 && ff == 1024 && !c->pose;   // tall chain kernels: face model only (built even under A2P_CHAIN_V=1: the switch is run-time)
  auto pf = [&](int l) { return "seqTransDecoder.stack." + std::to_string(l) + "."; };
  auto add_pre = [&](std::vector<ChainPackDesc>& v, int l) {  // [Q|K] then V of layer l's self attention
    const Buf& inw = c->wt.at(pf(l) + "self_attn.in_proj_weight");
    pk_gemm(v, inw.p, d, 2 * d, d, 1);
    pk_gemm(v, c->offT(inw, (int64_t)2 * d * d), d, d, d, 1);
  };
  for (int l = 0; l < L; ++l) {
    std::vector<ChainPackDesc> v;
    add_pre(v, l);
    CHK(chain_pack(c, ch_index(l, CH_PRE), v, {{W32(c, pf(l) + "self_attn.in_proj_bias"), 3 * d}}, s));
    auto mid = [&](int kind, const std::string& done, const std::string& next) -> int {
      std::vector<ChainPackDesc> m;
      pk_gemm(m, c->wt.at(pf(l) + done + ".out_proj.weight").p, d, d, d, 0, d / 128);
      pk_gemm(m, c->wt.at(pf(l) + next + ".in_proj_weight").p, d, d, d, 1);  // rows [0, d): the query projection
      return chain_pack(c, ch_index(l, kind), m, {{W32(c, pf(l) + next + ".in_proj_bias"), d}}, s);
    };
    CHK(mid(CH_MID, "self_attn", "multihead_attn"));
    if (c->pose) CHK(mid(CH_MID2, "multihead_attn", "multihead_attn2"));
    std::vector<ChainPackDesc> q;
    pk_gemm(q, c->wt.at(pf(l) + (c->pose ? "multihead_attn2" : "multihead_attn") + ".out_proj.weight").p, d, d, d, 0, d / 128);
    const Buf& w1 = c->wt.at(pf(l) + "linear1.weight");
    const Buf& w2 = c->wt.at(pf(l) + "linear2.weight");
    auto lin1 = [&](std::vector<ChainPackDesc>& v, int h) {
      for (int ks = 0; ks < d / 64; ++ks) v.push_back({reinterpret_cast<const h16_t*>(w1.p), d, h * 128, ks * 64, ff, 0});
    };
    auto lin2 = [&](std::vector<ChainPackDesc>& v, int h) {
      for (int ks = 0; ks < 2; ++ks)   // k-major over the d/128 output tiles (chain_body::gemm_group)
        for (int t = 0; t < d / 128; ++t) v.push_back({reinterpret_cast<const h16_t*>(w2.p), ff, t * 128, h * 128 + ks * 64, d, 0});
    };
    const int FTn = ff / 128;
    for (int h = 0; h < FTn; ++h) { lin1(q, h); lin2(q, h); }
    std::vector<std::pair<const float*, int>> aux = {{W32(c, pf(l) + "linear1.bias"), ff}};
    if (l + 1 < L) {
      add_pre(q, l + 1);
      aux.push_back({W32(c, pf(l + 1) + "self_attn.in_proj_bias"), 3 * d});
    } else if (!c->pose && !c->tail32) {  // A2P_TAIL16: final_layer of the face model rides on the last stream (16-bit operands)
      pk_gemm(q, c->wt.at("final_layer.weight").p, d, c->C, d);
      aux.push_back({W32(c, "final_layer.bias"), c->C});
    } else if (!c->pose && c->tail_x3) {  // the tall last-layer POST kernel computes final_layer as a split-operand island (ChainP::fin_x3): its bias behind bias_1
      aux.push_back({W32(c, "final_layer.bias"), c->C});
    }
    CHK(chain_pack(c, ch_index(l, CH_POST), q, aux, s));
    if (v4 && l == 0 && c->tail_x3 && c->C == 256 && c->Cpad == 256) {
      // CHAIN_IN: input_projection as a split-operand island (split weight rows [W_hi | W_hi | W_lo], 256 k each: hi x W_hi, lo x W_hi, hi x W_lo over the four output
      // tiles, k-chunk-major) followed by layer 0's PRE work ([Q|K] and V in groups of four tiles, as behind a POST kernel)
      std::vector<ChainPackDesc> in4;
      const void* wi = c->wt.at("input_projection.weight").p;
      pk4_gemm(in4, wi, 3 * c->Cpad, d, 0, c->Cpad, 0, 4);
      pk4_gemm(in4, wi, 3 * c->Cpad, d, 0, c->Cpad, 0, 4);
      pk4_gemm(in4, wi, 3 * c->Cpad, d, 2 * c->Cpad, c->Cpad, 0, 4);
      const Buf& inw = c->wt.at(pf(0) + "self_attn.in_proj_weight");
      pk4_gemm(in4, inw.p, d, 2 * d, 0, d, 1, 4);
      pk4_gemm(in4, c->offT(inw, (int64_t)2 * d * d), d, d, 0, d, 1, 4);
      CHK(chain4_pack(c, c->ch_stream4_in, in4, s));
      CHK(buf_alloc(c->ch_aux_in, 2560 * 4 + 1024));   // [input_projection.bias 512 | 512 unused | in_proj_bias 1536]: the offsets of a POST kernel's aux block
      HIPCHK(hipMemsetAsync(c->ch_aux_in.p, 0, 2560 * 4 + 1024, s));
      HIPCHK(hipMemcpyAsync(c->ch_aux_in.f(), W32(c, "input_projection.bias"), (size_t)d * 4, hipMemcpyDeviceToDevice, s));
      HIPCHK(hipMemcpyAsync(c->ch_aux_in.f() + 1024, W32(c, pf(0) + "self_attn.in_proj_bias"), (size_t)3 * d * 4, hipMemcpyDeviceToDevice, s));
      HIPCHK(hipStreamSynchronize(s));
    }
    if (v4) {   // the same chains for kernels_chain4.h (MID; POST of every layer that has a successor)
      std::vector<ChainPackDesc> m4;
      pk4_gemm(m4, c->wt.at(pf(l) + "self_attn.out_proj.weight").p, d, d, 0, d, 0, 4);
      pk4_gemm(m4, c->wt.at(pf(l) + "multihead_attn.in_proj_weight").p, d, d, 0, d, 1, 4);   // rows [0, d): the query projection, one group of four tiles
      CHK(chain4_pack(c, c->ch_stream4[ch_index(l, CH_MID)], m4, s));
      if (l + 1 < L || c->tail32) {   // (the last layer too, unless final_layer is fused into its POST kernel: A2P_TAIL16)
        for (int hc = 128; hc <= 256; hc += 128) {   // hidden chunk of the feed-forward block: 128 (80-row panels) | 256 (<= 64 rows): Chain4Lds::HC
          // two consumption orders of the feed-forward block (CHAIN4_FFN_PIPE, kernels_chain4.h): waves 4-7 (q4) linear1(h) linear2(h) per chunk; waves 0-3 (q4e)
          // run their linear2 partial one chunk late: linear1(0) | linear1(1) linear2(0) | linear1(2) linear2(1) | ... | linear2(last)
          std::vector<ChainPackDesc> q4, q4e;
          pk4_gemm(q4, c->wt.at(pf(l) + "multihead_attn.out_proj.weight").p, d, d, 0, d, 0, 4);
          q4e = q4;
          const int nh = hc / 128, nc = ff / hc;
          auto lin1_4 = [&](std::vector<ChainPackDesc>& v, int h) { pk4_gemm(v, w1.p, d, ff, 0, d, 0, nh, h * nh, nh); };   // linear1, hidden columns [hc h, hc h + hc): nh tiles, k-chunk-major
          auto lin2_4 = [&](std::vector<ChainPackDesc>& v, int h) { pk4_gemm(v, w2.p, ff, d, h * hc, hc, 0, 4); };          // linear2 partial over that hidden chunk, 4 output tiles
          for (int h = 0; h < nc; ++h) { lin1_4(q4, h); lin2_4(q4, h); }
          if (CHAIN4_FFN_PIPE) {
            for (int i = 0; i <= nc; ++i) {
              if (i < nc) lin1_4(q4e, i);
              if (i >= 1) lin2_4(q4e, i - 1);
            }
          } else {
            for (int h = 0; h < nc; ++h) { lin1_4(q4e, h); lin2_4(q4e, h); }
          }
          for (std::vector<ChainPackDesc>* v : {&q4, &q4e}) {
            if (l + 1 < L) {
              const Buf& inw = c->wt.at(pf(l + 1) + "self_attn.in_proj_weight");
              pk4_gemm(*v, inw.p, d, 2 * d, 0, d, 1, 4);                  // [Q|K] of the next layer, in groups of four tiles
              pk4_gemm(*v, c->offT(inw, (int64_t)2 * d * d), d, d, 0, d, 1, 4);   // V
            } else if (c->tail_x3 && !c->pose && c->C == 256) {
              // final_layer as a split-operand island inside the last POST kernel (chain4_kernel<.., 2>): the split weight rows are [W_hi | W_hi | W_lo] (make_wt3),
              // consumed as hi x W_hi, lo x W_hi, hi x W_lo: three 256 x 512 GEMMs in pairs of tiles
              const void* wf = c->wt.at("final_layer.weight").p;
              pk4_gemm(*v, wf, 3 * d, c->C, 0, d, 0, 2);
              pk4_gemm(*v, wf, 3 * d, c->C, 0, d, 0, 2);
              pk4_gemm(*v, wf, 3 * d, c->C, 2 * d, d, 0, 2);
            }
          }
          CHK(chain4_pack(c, hc == 128 ? c->ch_stream4[ch_index(l, CH_POST)] : c->ch_stream4w[l], q4, s, &q4e));
        }
      }
    }
    if (c->pose) {   // CHAIN_MIDPOST: out_proj of the audio cross attention | query projection of multihead_attn2 (k-major group: its
      // tiles stay in registers until the panel can be overwritten) | everything of the POST stream
      std::vector<ChainPackDesc> mp;
      pk_gemm(mp, c->wt.at(pf(l) + "multihead_attn.out_proj.weight").p, d, d, d, 0, d / 128);
      pk_gemm(mp, c->wt.at(pf(l) + "multihead_attn2.in_proj_weight").p, d, d, d, 0, d / 128);
      mp.insert(mp.end(), q.begin(), q.end());
      CHK(chain_pack(c, ch_index(l, CH_MIDPOST), mp, aux, s));
    }
  }
  return 0;
}

static void chain_base(a2p_ctx* c, ChainP& p, int N, int T, int idx, int aux_floats) {
  memset(&p, 0, sizeof(p));
  p.M = N * T; p.rows_per_seq = T; p.x = c->x.f(); p.cst = reinterpret_cast<const f32x4*>(c->rope_cst.p); p.cs_npos = c->rope_npos;
  p.ain = reinterpret_cast<const h16_t*>(c->ao.p); p.ld_ain = c->d;
  p.stream = reinterpret_cast<const h16_t*>(c->ch_stream[(size_t)(c->ch_nw == 8) * c->L * CH_KINDS + idx].p);
  p.stream4 = c->ch_stream4.empty() ? nullptr : reinterpret_cast<const h16_t*>(c->ch_stream4[idx].p);
  p.stream4w = (c->ch_stream4w.empty() || idx % CH_KINDS != CH_POST) ? nullptr : reinterpret_cast<const h16_t*>(c->ch_stream4w[idx / CH_KINDS].p);
  p.aux = c->ch_aux[idx].f(); p.aux_kb = (aux_floats + 255) / 256;
  if (c->clk.p) {  // A2P_CHAIN_CLK=1: every chain launch of a forward gets its own 8 x 4 slot (a2p_debug_read "clk")
#ifdef A2P_STAMPS   // diagnostic build (scratch/phase_probe.py): launch A2P_STAMP_LAUNCH of every forward writes its phase stamps behind the clk slots
    static const int sel = getenv("A2P_STAMP_LAUNCH") ? atoi(getenv("A2P_STAMP_LAUNCH")) : 4;
    p.fin_out = reinterpret_cast<float*>(reinterpret_cast<unsigned long long*>(c->clk.p) + 64 * 32 + ((int)c->clk_turn == sel ? 0 : 64));
#endif
    p.clk = reinterpret_cast<unsigned long long*>(c->clk.p) + (size_t)(c->clk_turn++ % 64) * 32;
  }
}

// outputs of the "pre" work of decoder layer l: norm1 -> rotary -> [Q|K], V^T
static void chain_set_pre(a2p_ctx* c, ChainP& p, int l, int T) {
  const int d = c->d;
  const std::string pf = "seqTransDecoder.stack." + std::to_string(l) + ".";
  p.lnB_g = W32(c, pf + "norm1.weight"); p.lnB_b = W32(c, pf + "norm1.bias");
  p.qk_out = reinterpret_cast<h16_t*>(c->qk.p); p.ld_qk = 2 * d;
  p.vt_out = reinterpret_cast<h16_t*>(c->vt.p);
  const int Tld = rup(T, 64);
  p.vt_seq_stride = (int64_t)d * Tld; p.ld_vt = Tld;
}

static void chain_set_out_proj(a2p_ctx* c, ChainP& p, const std::string& attn, const FilmRef& fr, int film_idx) {
  const int d = c->d;
  p.bias_o = W32(c, attn + ".out_proj.bias");
  if (fr.base) {
    p.film_o = fr.base + (int64_t)film_idx * 2 * d; p.film_seq_stride = fr.seq_stride; p.film_shift_off = d;
  }
}

// Workgroup shape of the chain kernels for a forward of `rows` rows: 4 waves (one 512-register wave per SIMD) or 8 (two
// 256-register waves per SIMD, panels of at most 48 / 64 rows).  Both give bit-identical results for both model widths
// (kernels_chain.h: one shared LayerNorm reduction tree, no floating-point contraction; tests/test_hip_round2.py).
// Which one is faster depends on the BOX, not on the code: on most MI355X boxes NW=4 leads by 3-5 % at B=8, on a sizeable
// minority the 512-register kernels run 35 % slower inside the step (not in isolation) and NW=8 leads by 19 % (docs/lab_notebook_r1_r4.md
// section 6).  So it is measured in situ: forwards 1..4 of a given size alternate the two shapes with an event pair around
// the decoder stack (forward 0 is warm-up), forward 5 picks the faster average and the choice sticks.  A2P_CHAIN_NW=4|8 forces.
// Round 4: the in-situ measurement is OPT-IN (A2P_CHAIN_TUNE=1).  The default is the 8-wave shape everywhere: deterministic per
// process and per rank (round 3's tuner timed 4 forwards per size and box: run-to-run noise decided close calls, and the ranks of
// one job could disagree), immune to the slow-box regime, and the only shape with mixed panel heights at B=32; what it gives up is
// the 4-wave shape's 3-5 % at B=8 on the boxes where that shape runs well.
static const int kTuneForwards = 5;
static int chain_pick_nw(a2p_ctx* c, int64_t rows, hipEvent_t* e0, hipEvent_t* e1) {
  *e0 = *e1 = nullptr;
  if (c->opt.chain_nw) return c->opt.chain_nw == 8 ? 8 : 4;
  if (!c->opt.chain_tune) {
    const int mt = c->opt.chain_mt;
    if (mt == 5 && c->d == 512 && c->opt.chain_v != 1 && !c->ch_stream4.empty() && c->ch_stream4[ch_index(0, CH_MID)].p) return 8;   // 80 rows: the tall kernels (8 waves) have it
    return (mt && ((mt > 4 && c->d == 512) || mt > 5)) ? 4 : 8;   // forced panel heights the 8-wave kernels do not have
  }
  if (c->opt.chain_mt) {  // a forced panel height the 8-wave kernels do not have
    const int mt = c->opt.chain_mt;
    if ((mt > 4 && c->d == 512) || mt > 5) return 4;
  }
  auto& t = c->ch_tune[rows];
  if (t.choice) return t.choice;
  const int call = t.calls++;
  if (call < kTuneForwards) {
    const int nw = (call & 1) ? 8 : 4;
    if (call >= 1 && hipEventCreate(e0) == hipSuccess && hipEventCreate(e1) == hipSuccess) t.samples.emplace_back(nw, *e0, *e1);
    else *e0 = *e1 = nullptr;
    return nw;
  }
  double sum[2] = {0, 0};
  int cnt[2] = {0, 0};
  for (auto& sm : t.samples) {
    float ms = 0.f;
    if (hipEventSynchronize(std::get<2>(sm)) == hipSuccess && hipEventElapsedTime(&ms, std::get<1>(sm), std::get<2>(sm)) == hipSuccess) {
      sum[std::get<0>(sm) == 8] += ms;
      ++cnt[std::get<0>(sm) == 8];
    }
    (void)hipEventDestroy(std::get<1>(sm));
    (void)hipEventDestroy(std::get<2>(sm));
  }
  t.samples.clear();
  t.choice = (cnt[0] && cnt[1] && sum[1] / cnt[1] < sum[0] / cnt[0]) ? 8 : 4;
  if (c->opt.tune_verbose)
    fprintf(stderr, "[a2p] chain workgroup shape for %lld rows: NW=4 %.3f ms, NW=8 %.3f ms -> %d\n", (long long)rows,
            cnt[0] ? sum[0] / cnt[0] : -1.0, cnt[1] ? sum[1] / cnt[1] : -1.0, t.choice);
  return t.choice;
}

// Chain kernel FAMILY of a forward, per chain: kernels_chain.h (1) or kernels_chain4.h (4) for the MID kernels and for the POST
// kernels.  The two families produce the same bits (tests/test_hip_round5.py), so the choice is invisible in the results.  What the
// boxes of rounds 5 and 6 showed (profiles/r05_boxes.md, r05_tall_chain_same_box_*.txt, r05_family_calibration.txt): the tall MID kernel
// wins on EVERY box (fast type 26.7 vs 30.9 us at B=8; slow type -- the boxes that also run the memory-heavy PRE kernel 30 % slow --
// 27 vs 34, B=32 88 vs 106), so MID is tall by rule; the tall POST kernel wins on most boxes (B=8: 71 vs 83 us) and LOSES on the slow
// type (B=32: 330 vs 278 us), so POST alone is measured in situ.  Round 5 timed all four (MID, POST) combinations twice each and took
// the smallest MEAN: a 2 % difference decided by two samples, one of which carried a kernel's first launch -- round 6 saw it pick the
// gen-1 MID kernels on a fast box (family 14: 664 instead of 676 steps/s).  Now: forwards 0 and 1 of a size warm both candidates up
// (untimed), forwards 2..9 alternate them with an event pair around the decoder stack (four samples each), forward 10 compares the
// MINIMA (interference from other streams only ever adds time) and keeps gen 1 only if it leads by more than 1.5 %.
// A2P_CHAIN_V=1 | 4 forces one family for both chains, 41 the mixed choice (tall MID, gen-1 POST).
static const int kFamConfigs[2][2] = {{4, 4}, {4, 1}};
static const int kTuneForwards4 = 10;
static void chain_pick_family(a2p_ctx* c, int64_t rows, hipEvent_t* e0, hipEvent_t* e1) {
  *e0 = *e1 = nullptr;
  auto set = [&](int cfg) { c->ch_fam_mid = kFamConfigs[cfg][0]; c->ch_fam_post = kFamConfigs[cfg][1]; };
  if (c->opt.chain_v == 1 || c->ch_stream4.empty() || c->d != 512 || c->ch_nw != 8) { c->ch_fam_mid = c->ch_fam_post = 1; return; }
  if (c->opt.chain_v == 4) return set(0);
  if (c->opt.chain_v == 41) return set(1);   // (tests, A/B: what the calibration picks on the slow GPU type -- tall MID, gen-1 POST)
  if (c->ch_tune4.size() > 64 && !c->ch_tune4.count(rows)) {   // (variable-length clips: the table stays bounded; pending event pairs are released)
    for (auto& kv : c->ch_tune4)
      for (auto& sm : kv.second.samples) { (void)hipEventDestroy(std::get<1>(sm)); (void)hipEventDestroy(std::get<2>(sm)); }
    c->ch_tune4.clear();
  }
  auto& t = c->ch_tune4[rows];
  if (t.choice) return set(t.choice - 1);
  const int call = t.calls++;
  if (call < kTuneForwards4) {
    const int cfg = call & 1;
    if (call >= 2 && hipEventCreate(e0) == hipSuccess && hipEventCreate(e1) == hipSuccess) t.samples.emplace_back(cfg, *e0, *e1);
    else *e0 = *e1 = nullptr;
    return set(cfg);
  }
  double best_ms[2] = {1e30, 1e30};
  for (auto& sm : t.samples) {
    float ms = 0.f;
    if (hipEventSynchronize(std::get<2>(sm)) == hipSuccess && hipEventElapsedTime(&ms, std::get<1>(sm), std::get<2>(sm)) == hipSuccess)
      best_ms[std::get<0>(sm)] = std::min(best_ms[std::get<0>(sm)], (double)ms);
    (void)hipEventDestroy(std::get<1>(sm));
    (void)hipEventDestroy(std::get<2>(sm));
  }
  t.samples.clear();
  const int best = (best_ms[1] < 1e29 && best_ms[0] < 1e29 && best_ms[1] < 0.985 * best_ms[0]) ? 1 : 0;   // (no measurement: the tall family)
  t.choice = best + 1;
  if (c->opt.tune_verbose)
    fprintf(stderr, "[a2p] chain kernel family of the POST chain for %lld rows (MID is tall by rule), fastest decoder stack of 4, ms: tall %.3f  gen 1 %.3f -> (%d,%d)\n",
            (long long)rows, best_ms[0], best_ms[1], kFamConfigs[best][0], kFamConfigs[best][1]);
  set(best);
}

// Tall chain kernels (kernels_chain4.h): 64 / 80-row panels, weights straight into registers, epilogue operands from LDS.
// Contract: face model (d = 512), MID or POST-with-successor, FiLM present, frame count a multiple of 8 and >= the panel height.
// Rule: forwards of >= 16 sequences (B >= 8 under guidance is 16 sequences of 600 frames = 37.5 rows per CU: the 48-row panels of
// kernels_chain.h are one round there; from 75 rows per CU on -- B = 16 -- an 80-row panel is one round where 48-row panels are two).
// final_layer inside the last layer's POST kernel (chain4_kernel<MT, CHAIN_POST, 2>) asked for and possible: split-operand islands (the default), 256 output features.
// The diagnostic build keeps its stamps in ChainP::fin_out.
static bool chain4_final_ok(const a2p_ctx* c, int mode, const ChainP& p) {
#ifdef A2P_STAMPS
  return false;
#else
  return mode == CHAIN_POST && p.has_next == 0 && p.fin_x3 && c->tail_x3 && !c->pose && c->C == 256 && !c->opt.no_fused_final;
#endif
}
static bool chain4_wanted(const a2p_ctx* c, int mode, const ChainP& p) {
  if (c->opt.chain_v == 1 || p.stream4 == nullptr || c->d != 512 || c->ch_nw != 8) return false;
  if (!(mode == CHAIN_MID || (mode == CHAIN_POST && p.has_next <= 1))) return false;   // (has_next == 2: final_layer fused, kernels_chain.h only)
  if (p.film_o == nullptr || (mode == CHAIN_POST && p.film_f == nullptr)) return false;
  if ((p.rows_per_seq & 7) || p.rows_per_seq < 80) return false;
  // The LAST layer's POST kernel is tall whatever the calibration says about the other seven: with final_layer inside it replaces three launches and 49 MB of traffic
  // (B=8: 53 us against 48 + 12.5 + 29), which no box type's gen-1 lead (at most ~20 % of one POST launch on the slow type) outweighs.  Same bits either way.
  if (chain4_final_ok(c, mode, p)) return true;
  return (mode == CHAIN_MID ? c->ch_fam_mid : c->ch_fam_post) == 4;   // the family of this forward's chain (chain_pick_family: forced, or measured on this box)
}
static int chain4_pick_mt(const a2p_ctx* c, int M) {
  int mt = c->opt.chain_mt;
  if (mt < 3 || mt > 5) {   // fewest rounds over the 256 CUs, then the shorter panel
    int best = 1 << 30;
    for (int m = 3; m <= 5; ++m) {
      const int blocks = (M + 16 * m - 1) / (16 * m);
      const int cost = ((blocks + 255) / 256) * (16 + 4 * m);
      if (cost < best) { best = cost; mt = m; }
    }
  }
  return mt;
}
static bool chain4_final_fused(const a2p_ctx* c, int mode, const ChainP& p) { return chain4_final_ok(c, mode, p) && chain4_wanted(c, mode, p); }
static int launch_chain4(a2p_ctx* c, int mode, const ChainP& p0, hipStream_t s) {
  ChainP p = p0;
  const int mt = chain4_pick_mt(c, p.M);
  const bool fin = chain4_final_fused(c, mode, p);
  p.stream = (mode == CHAIN_POST && mt <= 4) ? p.stream4w : p.stream4;   // POST: the stream of the panel height's hidden-chunk width (Chain4Lds::HC)
  const int grid = (p.M + 16 * mt - 1) / (16 * mt);
  ++c->ch4_launches;
  if (fin) ++c->fin_fused_launches;
  KernelTimer kt(c, A2P_KERNEL_CHAIN, mode == CHAIN_MID ? A2P_KERNEL_CHAIN_MID : A2P_KERNEL_CHAIN_POST);
#define A2P_CHAIN4(MT)                                                                          \
  do {                                                                                          \
    if (mode == CHAIN_MID) A2P_LAUNCH(kt, (chain4_kernel<MT, CHAIN_MID>), grid, 512, s, p);     \
    else if (p.has_next) A2P_LAUNCH(kt, (chain4_kernel<MT, CHAIN_POST>), grid, 512, s, p);      \
    else A2P_LAUNCH(kt, (chain4_kernel<MT, CHAIN_POST, 1>), grid, 512, s, p);                   \
  } while (0)
  if (fin) {
    if (mt == 3) A2P_LAUNCH(kt, (chain4_kernel<3, CHAIN_POST, 2>), grid, 512, s, p);
    else if (mt == 4) A2P_LAUNCH(kt, (chain4_kernel<4, CHAIN_POST, 2>), grid, 512, s, p);
    else A2P_LAUNCH(kt, (chain4_kernel<5, CHAIN_POST, 2>), grid, 512, s, p);
  } else if (mt == 3) A2P_CHAIN4(3);
  else if (mt == 4) A2P_CHAIN4(4);
  else A2P_CHAIN4(5);
#undef A2P_CHAIN4
  HIPCHK(hipGetLastError());
  return 0;
}

// input_projection + layer 0's PRE work as ONE tall kernel (chain4_kernel<MT, CHAIN_IN>) instead of pack_input_split3 + gemm_kernel + the gen-1 PRE kernel.
// Face model with split-operand islands, 256 input features, the 8-wave shape, clips of >= 80 frames (a multiple of 8), and one set of rows for the whole pass
// (no guidance, or guidance with the shared layer-0 half).  A2P_CHAIN_V=1 keeps the gen-1 launches.
static bool chain_in_wanted(const a2p_ctx* c, int T, bool rows_shared) {
#ifdef A2P_STAMPS
  return false;
#else
  return c->ch_stream4_in.p && c->opt.chain_v != 1 && !c->opt.no_fused_in && c->d == 512 && !c->pose && c->tail_x3 && c->C == 256 && c->ch_nw == 8 &&
         (T & 7) == 0 && T >= 80 && rows_shared;
#endif
}
static int launch_chain_in(a2p_ctx* c, const ChainP& p0, hipStream_t s) {
  ChainP p = p0;
  const int mt = chain4_pick_mt(c, p.M);
  p.stream = reinterpret_cast<const h16_t*>(c->ch_stream4_in.p);
  const int grid = (p.M + 16 * mt - 1) / (16 * mt);
  ++c->ch4_launches;
  ++c->in4_launches;
  KernelTimer kt(c, A2P_KERNEL_CHAIN, A2P_KERNEL_CHAIN_PRE);
  if (mt == 3) A2P_LAUNCH(kt, (chain4_kernel<3, CHAIN_IN>), grid, 512, s, p);
  else if (mt == 4) A2P_LAUNCH(kt, (chain4_kernel<4, CHAIN_IN>), grid, 512, s, p);
  else A2P_LAUNCH(kt, (chain4_kernel<5, CHAIN_IN>), grid, 512, s, p);
  HIPCHK(hipGetLastError());
  return 0;
}

static int launch_chain(a2p_ctx* c, int mode, const ChainP& p, hipStream_t s) {
  if (chain4_wanted(c, mode, p)) return launch_chain4(c, mode, p, s);
  const bool env_mt = c->opt.chain_mt != 0;  // tuning / test override of the panel height (rows = 16 * MT)
  int mt = c->opt.chain_mt;
  // panel heights instantiated per width (LDS: the [16*MT][d] bf16 panel + hidden chunk + >= 3 ring slots must fit 160 KiB)
  static const int kMt512[] = {4, 3, 2}, kMt256[] = {6, 5, 4, 3, 2};
  static const int kMt512w8[] = {4, 3, 2}, kMt256w8[] = {5, 4, 3, 2};  // 8 waves: 256 registers per wave bound the panel height (d = 256, 96 rows: 46 spilled)
  const bool w8 = c->ch_nw == 8;
  const int* cands = c->d == 512 ? (w8 ? kMt512w8 : kMt512) : (w8 ? kMt256w8 : kMt256);
  const int ncand = c->d == 512 ? 3 : (w8 ? 4 : 5);
  bool ok = false;
  for (int i = 0; i < ncand; ++i) ok = ok || cands[i] == mt;
  if (!ok) {  // fewest rounds over the 256 CUs, then the cheaper (shorter) panel: every panel streams all weights once.
    // d = 512: the 64-row panels are left to A2P_CHAIN_MT -- their POST kernels spill (20 B / 340 B of scratch per lane with
    // 4 / 8 waves) and measure slower at every size tried (B=32: 133 vs 141 steps/s with 4 waves, 143 vs 148 with 8)
    int best = 1 << 30;
    for (int i = 0; i < ncand; ++i) {
      if (c->d == 512 && cands[i] == 4) continue;
      const int blocks = (p.M + 16 * cands[i] - 1) / (16 * cands[i]);
      const int cost = ((blocks + 255) / 256) * (16 + 4 * cands[i]);
      if (cost < best) { best = cost; mt = cands[i]; }
    }
  }
  const int grid = (p.M + 16 * mt - 1) / (16 * mt);
  KernelTimer kt(c, A2P_KERNEL_CHAIN, mode == CHAIN_PRE ? A2P_KERNEL_CHAIN_PRE : mode == CHAIN_MID ? A2P_KERNEL_CHAIN_MID
                                       : mode == CHAIN_POST ? A2P_KERNEL_CHAIN_POST : A2P_KERNEL_CHAIN_MIDPOST);
  // d = 512, more than one round of 48-row panels with a mostly empty last round: 64-row panels for the first n_tall
  // workgroups so that the forward fits one round less (chain_kernel_mix; B=32: 96 x 64 + 672 x 48 rows = 3 full rounds
  // instead of 3.125 -> 4).  A2P_CHAIN_NO_MIX=1 keeps the uniform launch for A/B runs.
  // Only for the 8-wave shape: 64-row panels of 512-register waves live half in AGPRs and run 1.6-1.9x the 48-row time
  // (scratch/chain_bench, POST: 143 vs 89 us), with 8 waves 1.26-1.37x (102 vs 75 us) for 1.33x the rows.
  if (w8 && c->d == 512 && mt == 3 && !env_mt && grid > 256 && grid % 256 != 0 && !c->opt.chain_no_mix) {
    const int W = 256 * ((grid + 255) / 256 - 1);
    const int n_tall = (p.M - 48 * W + 15) / 16;
    if (n_tall > 0 && n_tall <= W) {
      ChainP q = p;
      q.n_tall = n_tall;
#define A2P_CHAIN_MIX(NW)                                                                                              \
  do {                                                                                                                 \
    if (mode == CHAIN_PRE) A2P_LAUNCH(kt, (chain_kernel_mix<512, 4, 3, CHAIN_PRE, NW>), W, 64 * NW, s, q);              \
    else if (mode == CHAIN_MID) A2P_LAUNCH(kt, (chain_kernel_mix<512, 4, 3, CHAIN_MID, NW>), W, 64 * NW, s, q);         \
    else A2P_LAUNCH(kt, (chain_kernel_mix<512, 4, 3, CHAIN_POST, NW>), W, 64 * NW, s, q);                               \
  } while (0)
      A2P_CHAIN_MIX(8);
#undef A2P_CHAIN_MIX
      HIPCHK(hipGetLastError());
      return 0;
    }
  }
#ifdef A2P_STAMPS
#define A2P_CHAIN_ABL 64
#else
#define A2P_CHAIN_ABL 0
#endif
#define A2P_CHAIN_W(D, MT, NW)                                                                                              \
  do {                                                                                                                      \
    if (mode == CHAIN_PRE) A2P_LAUNCH(kt, (chain_kernel<D, MT, CHAIN_PRE, A2P_CHAIN_ABL, NW>), grid, 64 * NW, s, p);        \
    else if (mode == CHAIN_MID) A2P_LAUNCH(kt, (chain_kernel<D, MT, CHAIN_MID, A2P_CHAIN_ABL, NW>), grid, 64 * NW, s, p);   \
    else A2P_LAUNCH(kt, (chain_kernel<D, MT, CHAIN_POST, A2P_CHAIN_ABL, NW>), grid, 64 * NW, s, p);                         \
  } while (0)
#define A2P_CHAIN(D, MT) A2P_CHAIN_W(D, MT, 4)
  if (mode == CHAIN_MIDPOST) {   // body model only (d = 256): MID2 | keyframe attention | POST in one kernel
    ARG(c->d == 256, "CHAIN_MIDPOST is instantiated for d = 256");
#define A2P_CHAIN_MP(MT, NW) A2P_LAUNCH(kt, (chain_kernel<256, MT, CHAIN_MIDPOST, A2P_CHAIN_ABL, NW>), grid, 64 * NW, s, p)
    if (w8) {
      if (mt == 2) A2P_CHAIN_MP(2, 8);
      else if (mt == 3) A2P_CHAIN_MP(3, 8);
      else if (mt == 4) A2P_CHAIN_MP(4, 8);
      else A2P_CHAIN_MP(5, 8);
    } else {
      if (mt == 2) A2P_CHAIN_MP(2, 4);
      else if (mt == 3) A2P_CHAIN_MP(3, 4);
      else if (mt == 4) A2P_CHAIN_MP(4, 4);
      else if (mt == 5) A2P_CHAIN_MP(5, 4);
      else A2P_CHAIN_MP(6, 4);
    }
#undef A2P_CHAIN_MP
    HIPCHK(hipGetLastError());
    return 0;
  }
  if (w8) {
    if (c->d == 512) {
      if (mt == 2) A2P_CHAIN_W(512, 2, 8);
      else if (mt == 3) A2P_CHAIN_W(512, 3, 8);
      else A2P_CHAIN_W(512, 4, 8);
    } else {
      if (mt == 2) A2P_CHAIN_W(256, 2, 8);
      else if (mt == 3) A2P_CHAIN_W(256, 3, 8);
      else if (mt == 4) A2P_CHAIN_W(256, 4, 8);
      else A2P_CHAIN_W(256, 5, 8);
    }
  } else if (c->d == 512) {
    if (mt == 2) A2P_CHAIN(512, 2);
    else if (mt == 3) A2P_CHAIN(512, 3);
    else A2P_CHAIN(512, 4);
  } else {
    if (mt == 2) A2P_CHAIN(256, 2);
    else if (mt == 3) A2P_CHAIN(256, 3);
    else if (mt == 4) A2P_CHAIN(256, 4);
    else if (mt == 5) A2P_CHAIN(256, 5);
    else A2P_CHAIN(256, 6);
  }
#undef A2P_CHAIN
#undef A2P_CHAIN_W
  HIPCHK(hipGetLastError());
  return 0;
}

static int launch_self_attention(a2p_ctx* c, int N, int T, hipStream_t s, bool ksplit = false) {
  const int d = c->d, Tld = rup(T, 64);
  AttnP a;
  memset(&a, 0, sizeof(a));
  a.Q = c->qk.p; a.q_seq_stride = (int64_t)T * 2 * d; a.ldq = 2 * d;
  a.K = c->offT(c->qk, d); a.k_slot_stride = (int64_t)T * 2 * d; a.ldk = 2 * d;
  a.VT = c->vt.p; a.vt_slot_stride = (int64_t)d * Tld; a.ldvt = Tld;
  a.O = c->ao.p; a.o_seq_stride = (int64_t)T * d; a.ldo = d;
  a.tail_mod = 1; a.Tq = T; a.S_main = T; a.S_tail = 0;
  a.scale_log2e = 1.4426950408889634f / sqrtf((float)c->DH);
  return launch_attn(c, a, N, A2P_KERNEL_ATTN_SELF, s, ksplit);
}

static int launch_cross_attention(a2p_ctx* c, int N, int T, const CrossKV& kv, hipStream_t s, bool ksplit = false) {
  const int d = c->d;
  AttnP a;
  memset(&a, 0, sizeof(a));
  a.Q = c->qk.p; a.q_seq_stride = (int64_t)T * d; a.ldq = d;
  a.K = kv.K; a.k_slot_stride = kv.k_slot_stride; a.ldk = kv.ldk;
  a.VT = kv.VT; a.vt_slot_stride = kv.vt_slot_stride; a.ldvt = kv.ldvt;
  a.O = c->ao.p; a.o_seq_stride = (int64_t)T * d; a.ldo = d;
  a.ktail = kv.ktail; a.vtail = kv.vtail; a.tail_sample_stride = kv.tail_sample_stride; a.tail_row_stride = kv.tail_row_stride;
  a.kv_slot = kv.slots; a.tail_mod = kv.tail_mod; a.Tq = T; a.S_main = kv.S_main; a.S_tail = kv.S_tail;
  a.slot_rule = kv.slot_rule; a.slot_b = kv.slot_b;
  a.kv_stream = kv.slots && !c->opt.kv_cached ? 1 : 0;
  a.scale_log2e = 1.4426950408889634f / sqrtf((float)c->DH);
  return launch_attn(c, a, N, A2P_KERNEL_ATTN_CROSS, s, ksplit);
}

// FiLMTransformerDecoderLayer.forward as PRE? | self attention | MID | cross attention | (MID | cross attention 2) | POST
// shared_half: classifier-free guidance, layer 0 -- both halves of the 2B sequences start from the same x (one input
// projection) and therefore share norm1 / Q,K,V / the self attention: those run on the first N/2 sequences only, and the MID
// kernel of the second half reads the first half's rows (ChainP::src_rows)
static int decoder_layer_chain(a2p_ctx* c, int l, int N, int T, const CrossKV& kv, const CrossKV* kv2, const FilmRef& fr, bool first,
                               bool has_next, hipStream_t s, hipEvent_t film_ready = nullptr, bool fuse_final = false,
                               bool shared_half = false, bool* fused_x3 = nullptr, const float* x_in_fused = nullptr) {
  const int d = c->d;
  const std::string pf = "seqTransDecoder.stack." + std::to_string(l) + ".";
  ChainP p;
  const int join_at = c->opt.side_join;  // diagnostic: where the main stream joins the side stream (1 PRE, 2 self attention, default MID)
  if (film_ready && join_at <= 1) HIPCHK(hipStreamWaitEvent(s, film_ready, 0));
  const int Nsa = shared_half ? N / 2 : N;
  const bool tiled = !c->opt.x_rowmajor;   // A/B switch: keep the residual stream row-major between chain kernels
  // shared_half: the input projection wrote the (N/2)*T rows both halves start from into c->hff (unused by the chain path, fp32
  // here); PRE reads them there, MID reads them there for BOTH halves and writes all N*T rows of c->x -- never in place: a
  // second-half workgroup may start after the first-half workgroup of the same source rows has stored its result
  float* x0 = shared_half ? c->hff.f() : nullptr;
  if (first) {
    chain_base(c, p, Nsa, T, ch_index(l, CH_PRE), 3 * d);
    chain_set_pre(c, p, l, T);
    if (x0) p.x = x0;
    if (x_in_fused) {   // input_projection inside the kernel: the rows it computes are stored (row-major) where the GEMM left them
      p.xin = x_in_fused; p.xin_C = c->C;
      p.aux = c->ch_aux_in.f(); p.aux_kb = 10;
      p.x_in_tiled = 0; p.x_out_tiled = 0;
      CHK(launch_chain_in(c, p, s));
    } else {
      CHK(launch_chain(c, CHAIN_PRE, p, s));
    }
  }
  if (film_ready && join_at == 2) HIPCHK(hipStreamWaitEvent(s, film_ready, 0));
  CHK(launch_self_attention(c, Nsa, T, s));
  if (film_ready && join_at >= 3) HIPCHK(hipStreamWaitEvent(s, film_ready, 0));  // FiLM / time-token K,V of this step (side stream)
  auto mid = [&](int kind, const std::string& attn_done, int film_idx, const std::string& norm) -> int {
    chain_base(c, p, N, T, ch_index(l, kind), d);
    chain_set_out_proj(c, p, pf + attn_done, fr, film_idx);
    p.lnA_g = W32(c, pf + norm + ".weight"); p.lnA_b = W32(c, pf + norm + ".bias");
    p.q_out = reinterpret_cast<h16_t*>(c->qk.p); p.ld_q = d;
    if (shared_half && kind == CH_MID) {
      p.src_rows = (N / 2) * T;
      p.xsrc = x0;
    }
    // residual layout (ChainP::x_in_tiled): row-major as the input projection (or the caller) left it for the first kernel that
    // touches the rows, tiled between chain kernels
    p.x_in_tiled = !(first && kind == CH_MID) && tiled;
    p.x_out_tiled = tiled;
    return launch_chain(c, CHAIN_MID, p, s);
  };
  CHK(mid(CH_MID, "self_attn", 0, "norm2"));
  CHK(launch_cross_attention(c, N, T, kv, s));
  // body model: MID2 | keyframe attention | POST as ONE kernel (kernels_chain.h CHAIN_MIDPOST) when the keyframes fit one 32-key chunk
  // (T <= 960 frames at the reference's keyframe step of 30); A2P_NO_FUSED_KF=1 keeps the three launches (A/B, tests)
  // The fused attention phase is written for 8 heads x 32 (the reference's pose configuration): wave w = head w, K / V^T fragments of
  // 32-wide heads.  Any other head geometry at d = 256 (the reference's argparse default is --heads 4, i.e. head_dim 64) takes the
  // three launches.  A 16-row tile of the panel may straddle TWO sequences (the kernel keeps both key sets), not three: T >= 16.
  const bool fuse_kf = kv2 && d == 256 && c->H == 8 && c->DH == 32 && T >= 16 && kv2->S_main >= 1 && kv2->S_main <= 32 &&
                       kv2->S_tail == 0 && !c->opt.no_fused_kf && !fuse_final;
  if (kv2 && !fuse_kf) {
    CHK(mid(CH_MID2, "multihead_attn", 1, "norm2a"));
    CHK(launch_cross_attention(c, N, T, *kv2, s));
  }
  chain_base(c, p, N, T, ch_index(l, fuse_kf ? CH_MIDPOST : CH_POST), c->ff + (has_next ? 3 * d : 0));
  if (fuse_kf) {
    chain_set_out_proj(c, p, pf + "multihead_attn", fr, 1);                                      // first sublayer: the audio cross attention's output
    p.lnA_g = W32(c, pf + "norm2a.weight"); p.lnA_b = W32(c, pf + "norm2a.bias");
    p.bias_q2 = W32(c, pf + "multihead_attn2.in_proj_bias");
    p.bias_o2 = W32(c, pf + "multihead_attn2.out_proj.bias");
    if (fr.base) p.film_o2 = fr.base + (int64_t)3 * 2 * d;
    p.lnC_g = W32(c, pf + "norm3.weight"); p.lnC_b = W32(c, pf + "norm3.bias");
    p.k2 = reinterpret_cast<const h16_t*>(kv2->K); p.k2_slot_stride = kv2->k_slot_stride; p.ld_k2 = kv2->ldk;
    p.vt2 = reinterpret_cast<const h16_t*>(kv2->VT); p.vt2_slot_stride = kv2->vt_slot_stride; p.ld_vt2 = kv2->ldvt;
    p.kv2_slots = kv2->slots; p.kv2_rule = kv2->slot_rule; p.kv2_b = kv2->slot_b; p.n_key2 = kv2->S_main;
    p.scale2 = 1.0f / sqrtf((float)c->DH);
    p.stat_max = c->nonfinite.p ? reinterpret_cast<int*>(c->nonfinite.p) + 1 : nullptr;
  } else {
    chain_set_out_proj(c, p, pf + (kv2 ? "multihead_attn2" : "multihead_attn"), fr, kv2 ? 3 : 1);
    p.lnA_g = W32(c, pf + "norm3.weight"); p.lnA_b = W32(c, pf + "norm3.bias");
  }
  p.bias_2 = W32(c, pf + "linear2.bias");
  if (fr.base) p.film_f = fr.base + (int64_t)2 * 2 * d;
  p.has_next = has_next ? 1 : 0;
  p.x_in_tiled = tiled;
  p.x_out_tiled = tiled && has_next;   // the last layer's rows go back to the caller / the pose tail row-major
  if (has_next) chain_set_pre(c, p, l + 1, T);
  if (!has_next && fuse_final) {  // model output rows straight from the last POST kernel (c->x is NOT updated)
    p.has_next = 2; p.fin_out = c->mo.f(); p.ld_fin = c->C; p.fin_n = c->C;
    p.aux_kb = (c->ff + c->C + 255) / 256;
  }
  if (!has_next && fused_x3 && !fuse_kf && !fuse_final) {   // the caller wants the MODEL OUTPUT: the tall last-layer kernel computes final_layer too where it can (c->x is NOT updated then)
    float* const fin_out0 = p.fin_out;   // (the diagnostic build keeps its stamp slots there)
    p.fin_x3 = 1; p.fin_out = c->mo.f(); p.ld_fin = c->C; p.fin_n = c->C;
    p.aux_kb = (c->ff + c->C + 255) / 256;
    *fused_x3 = chain4_final_fused(c, CHAIN_POST, p);
    if (!*fused_x3) { p.fin_x3 = 0; p.fin_out = fin_out0; p.aux_kb = (c->ff + 255) / 256; }
  }
  return launch_chain(c, fuse_kf ? CHAIN_MIDPOST : CHAIN_POST, p, s);
}

// FiLMTransformerDecoderLayer.forward (transformer_modules.py:178-217) on c->x
static int decoder_layer(a2p_ctx* c, int l, int N, int T, const CrossKV& kv, const CrossKV* kv2, const FilmRef& fr, hipStream_t s) {
  const std::string p = "seqTransDecoder.stack." + std::to_string(l) + ".";
  CHK(self_attn_block(c, p + "self_attn", p + "norm1", N, T, fr, 0, s));
  CHK(cross_attn_block(c, p + "multihead_attn", p + "norm2", N, T, kv, fr, 1, s));
  if (kv2) CHK(cross_attn_block(c, p + "multihead_attn2", p + "norm2a", N, T, *kv2, fr, 3, s));
  return ffn_block(c, p, p + "norm3", N * T, T, fr, 2, s);
}

// ------------------------------------------------------------------------------------------------
// small forwards (16-bit modes, below A2POpts::chain_rows = 1100 rows, face model): whole-K-resident small-tile GEMMs with the LayerNorm fused into the A
// load (kernels_small.h): 8 launches per decoder layer instead of 12
// ------------------------------------------------------------------------------------------------
static bool small_supported(const a2p_ctx* c) {
  return c->bf16 && c->d == 512 && c->ff == 1024 && !c->pose && !c->opt.no_small;
}

template <int K, int BN, int PRO, int EPI, int BM = 32>
static int launch_small(a2p_ctx* c, const SmallP& p, hipStream_t s) {
  KernelTimer kt(c, A2P_KERNEL_GEMM);
  dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM);
  A2P_LAUNCH(kt, (small_gemm_kernel<K, BN, PRO, EPI, BM>), grid, 256, s, p);
  HIPCHK(hipGetLastError());
  return 0;
}

static int decoder_layer_small(a2p_ctx* c, int l, int N, int T, const CrossKV& kv, const FilmRef& fr, hipStream_t s,
                               hipEvent_t film_ready = nullptr) {
  const int d = c->d, ff = c->ff, M = N * T, Tld = rup(T, 64);
  const std::string pf = "seqTransDecoder.stack." + std::to_string(l) + ".";
  SmallP base;
  memset(&base, 0, sizeof(base));
  base.M = M; base.rows_per_seq = T; base.cs = reinterpret_cast<const float2*>(c->rope_cs.p);
  auto ln = [&](SmallP& p, const std::string& norm) {
    p.x = c->x.f(); p.gamma = W32(c, pf + norm + ".weight"); p.beta = W32(c, pf + norm + ".bias");
  };
  auto film = [&](SmallP& p, int idx) {
    p.resid = c->x.f(); p.film = fr.base + (int64_t)idx * 2 * d; p.film_seq_stride = fr.seq_stride; p.film_shift_off = d;
  };
  {  // norm1 -> rotary -> [Q|K] ; norm1 -> V^T : one launch over the 3d output columns
    SmallP p = base;
    ln(p, "norm1");
    p.n_rope = 2 * d; p.W = reinterpret_cast<const h16_t*>(c->wt.at(pf + "self_attn.in_proj_weight").p); p.bias = W32(c, pf + "self_attn.in_proj_bias");
    p.N = 3 * d; p.out = reinterpret_cast<h16_t*>(c->qk.p); p.ldo = 2 * d; p.n_store = 2 * d;
    p.out_t = reinterpret_cast<h16_t*>(c->vt.p); p.ld_t = Tld; p.t_seq_stride = (int64_t)d * Tld;
    CHK((launch_small<512, 64, 1, SMALL_STORE>(c, p, s)));
  }
  CHK(launch_self_attention(c, N, T, s, true));
  if (film_ready) HIPCHK(hipStreamWaitEvent(s, film_ready, 0));   // FiLM / time-token K,V of this step come from the side stream
  {  // out_proj + FiLM + residual
    SmallP p = base;
    p.a = reinterpret_cast<const h16_t*>(c->ao.p); p.lda = d; p.W = reinterpret_cast<const h16_t*>(c->wt.at(pf + "self_attn.out_proj.weight").p);
    p.bias = W32(c, pf + "self_attn.out_proj.bias"); p.N = d;
    film(p, 0);
    CHK((launch_small<512, 64, 0, SMALL_FILM_RES>(c, p, s)));
  }
  {  // norm2 -> rotary -> Q of the cross attention
    SmallP p = base;
    ln(p, "norm2");
    p.n_rope = d; p.W = reinterpret_cast<const h16_t*>(c->wt.at(pf + "multihead_attn.in_proj_weight").p); p.bias = W32(c, pf + "multihead_attn.in_proj_bias");
    p.N = d; p.out = reinterpret_cast<h16_t*>(c->qk.p); p.ldo = d; p.n_store = d;
    CHK((launch_small<512, 64, 1, SMALL_STORE>(c, p, s)));
  }
  CHK(launch_cross_attention(c, N, T, kv, s, true));
  {
    SmallP p = base;
    p.a = reinterpret_cast<const h16_t*>(c->ao.p); p.lda = d; p.W = reinterpret_cast<const h16_t*>(c->wt.at(pf + "multihead_attn.out_proj.weight").p);
    p.bias = W32(c, pf + "multihead_attn.out_proj.bias"); p.N = d;
    film(p, 1);
    CHK((launch_small<512, 64, 0, SMALL_FILM_RES>(c, p, s)));
  }
  {  // norm3 -> linear1 + GELU
    SmallP p = base;
    ln(p, "norm3");
    p.n_rope = 0; p.W = reinterpret_cast<const h16_t*>(c->wt.at(pf + "linear1.weight").p); p.bias = W32(c, pf + "linear1.bias");
    p.N = ff; p.out = reinterpret_cast<h16_t*>(c->hff.p); p.ldo = ff; p.n_store = ff; p.gelu = 1;
    CHK((launch_small<512, 64, 1, SMALL_STORE>(c, p, s)));
  }
  {  // linear2 + FiLM + residual (K = ff)
    SmallP p = base;
    p.a = reinterpret_cast<const h16_t*>(c->hff.p); p.lda = ff; p.W = reinterpret_cast<const h16_t*>(c->wt.at(pf + "linear2.weight").p);
    p.bias = W32(c, pf + "linear2.bias"); p.N = d;
    film(p, 2);
    CHK((launch_small<1024, 32, 0, SMALL_FILM_RES>(c, p, s)));
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// hoisted conditioning
// ------------------------------------------------------------------------------------------------
// K/V slot of every sequence of a pass: slot 1 + b = sample b's conditioning, slot 0 = the batch-invariant unconditional branch
__global__ void slot_tables_kernel(int* cond, int* unc, int* cfg, int B) {
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    cond[b] = 1 + b;
    unc[b] = 0;
    cfg[b] = 1 + b;      // classifier-free guidance: first half conditional ...
    cfg[B + b] = 0;      // ... second half unconditional
  }
}

extern "C" int a2p_prepare_cond(a2p_ctx* c, const float* cond_embed, int32_t B, int32_t S0, const float* keyframes,
                                const uint8_t* key_mask, int32_t n_key, int32_t T, void* stream) {
  ARG(c && cond_embed, "null argument");
  if (!c->finalized) {
    set_err("a2p_prepare_cond before a2p_finalize_weights");
    return A2P_ERR_STATE;
  }
  ARG(B >= 1 && B <= c->Bmax, "batch %d exceeds max_batch %d", B, c->Bmax);
  ARG(S0 >= 1 && S0 <= c->S0max, "n_tok %d exceeds emb_len %d", S0, c->S0max);
  ARG(T >= 1 && T <= c->Tmax, "frames %d outside [1, %d]", T, c->Tmax);
  if (c->pose) ARG(keyframes && n_key >= 1 && n_key <= c->KFmax, "pose model needs keyframes (1..%d)", c->KFmax);
  hipStream_t s = (hipStream_t)stream;
  const int d = c->d, M = B * S0;
  c->prepared = false;
  // cond_projection (model/diffusion.py:372)
  CHK(launch_cast(c, cond_embed, c->Fc, c->ce_pack.p, c->FcPad, M, c->Fc, c->FcPad, nullptr, s));
  GemmP pc = gemm_base(c->ce_pack.p, c->FcPad, c->wt.at("cond_projection.weight").p, c->FcPad, W32(c, "cond_projection.bias"),
                       c->x.p, d, M, d, c->FcPad);
  pc.out_f32 = 1;
  CHK(launch_gemm(c, pc, s));
  if (!c->pose) {  // cond_encoder: 2 x TransformerEncoderLayerRotary (transformer_modules.py:68-102)
    FilmRef none;
    for (int i = 0; i < 2; ++i) {
      const std::string p = "cond_encoder." + std::to_string(i) + ".";
      CHK(self_attn_block(c, p + "self_attn", p + "norm1", B, S0, none, 0, s));
      CHK(ffn_block(c, p, p + "norm2", M, S0, none, 0, s));
    }
  }
  // pooled hidden (model/diffusion.py:380-381) -> hidden slot 1+b
  mean_tokens_kernel<<<dim3((d + 63) / 64, B), 1024, 0, s>>>(c->x.f(), c->pooled.f(), S0, d);
  CHK(launch_ln_rope(c, true, c->pooled.f(), d, W32(c, "non_attn_cond_projection.0.weight"), W32(c, "non_attn_cond_projection.0.bias"),
                     c->tmpa.p, nullptr, d, B, B, 0, s));
  CHK(launch_skinny(c->tmpa.f(), d, W32(c, "non_attn_cond_projection.1.weight"), d, W32(c, "non_attn_cond_projection.1.bias"),
                    c->tmpb.f(), d, B, d, d, ACT_SILU, s));
  CHK(launch_skinny(c->tmpb.f(), d, W32(c, "non_attn_cond_projection.3.weight"), d, W32(c, "non_attn_cond_projection.3.bias"),
                    c->hidden.f() + d, d, B, d, d, ACT_NONE, s));
  // norm_cond + rotary of the audio tokens, then K / V^T of every decoder layer into slots 1..B
  CHK(launch_ln_rope(c, false, c->x.f(), d, W32(c, "norm_cond.weight"), W32(c, "norm_cond.bias"), c->xn.p, c->xr.p, d, M, S0, 0, s));
  CHK(project_kv_all(c, c->xr.p, c->xn.p, M, S0, c->cak_wt, c->cak_b.f(), c->cav_wt, c->cav_b.f(), c->kc.p, c->Sld, c->vtc.p, 1, s));
  if (c->pose) {  // encode_keyframes (model/diffusion.py:315-336), conditional branch
    const int Mk = B * n_key;
    HIPCHK(hipMemsetAsync(c->cb[0].p, 0, c->cb[0].bytes, s));  // left-pad rows of the conv tail must be zero for this T
    CHK(launch_cast(c, keyframes, c->Kd, c->kf_pack.p, c->KdPad, Mk, c->Kd, c->KdPad, key_mask, s));
    GemmP pk = gemm_base(c->kf_pack.p, c->KdPad, c->wt.at("frame_cond_projection.weight").p,
                         c->KdPad, W32(c, "frame_cond_projection.bias"), c->kf_tok.p, d, Mk, d, c->KdPad);
    pk.out_f32 = 1;
    CHK(launch_gemm(c, pk, s));
    CHK(launch_ln_rope(c, false, c->kf_tok.f(), d, W32(c, "frame_norm_cond.weight"), W32(c, "frame_norm_cond.bias"), c->xn.p, c->xr.p,
                       d, Mk, n_key, 0, s));
    CHK(project_kv_all(c, c->xr.p, c->xn.p, Mk, n_key, c->ca2k_wt, c->ca2k_b.f(), c->ca2v_wt, c->ca2v_b.f(), c->k2c.p, 64, c->vt2c.p,
                       1, s));
  }
  // slot tables, written on the device: no pageable host buffer, hence no stream synchronisation in this entry point
  // (round 1 synchronised here once per clip; config 4 loops subjects and clips)
  slot_tables_kernel<<<1, 256, 0, s>>>((int*)c->slot_cond.p, (int*)c->slot_unc.p, (int*)c->slot_cfg.p, B);
  HIPCHK(hipGetLastError());
  c->pB = B; c->pS0 = S0; c->pT = T; c->pK = n_key;
  c->prepared = true;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// per-step time path (a15, a20, time-token K/V)
// ------------------------------------------------------------------------------------------------
static int time_path(a2p_ctx* c, const int64_t* t_orig, int N, const int* slots, hipStream_t s, bool embed = true) {
  const int d = c->d, B = c->pB, L = c->L, F = c->F;
  // time_embed -> time_mlp -> [to_time_cond ; to_time_tokens] depend on the timestep VALUE alone: a row of the table built at a2p_finalize_weights (same kernels, same bits)
  // where there is one and this call may read the caller's timestep tensor (not under graph replay); three launches otherwise
  const bool table = embed && c->tct_rows > 0 && c->opt.time_table > 0;
  if (!table) {
    if (embed) time_embed_kernel<<<(B * (d / 2) + 255) / 256, 256, 0, s>>>(t_orig, c->time_freq.f(), c->emb.f(), B, d / 2);
    CHK(launch_skinny(c->emb.f(), d, W32(c, "time_mlp.1.weight"), d, W32(c, "time_mlp.1.bias"), c->th.f(), 4 * d, B, 4 * d, d, ACT_MISH, s));
    CHK(launch_skinny(c->th.f(), 4 * d, c->tct_w.f(), 4 * d, c->tct_b.f(), c->tct.f(), 3 * d, B, 3 * d, 4 * d, ACT_NONE, s));
  }
  TPathP tp;
  memset(&tp, 0, sizeof(tp));
  tp.tct = table ? c->tct_table.f() : c->tct.f();
  tp.trow = table ? t_orig : nullptr; tp.trows = c->tct_rows; tp.err = reinterpret_cast<int*>(c->nonfinite.p); tp.hidden = c->hidden.f(); tp.slot = slots; tp.tvec = c->tvec.f(); tp.mt = c->mt.f();
  tp.gamma = W32(c, "norm_cond.weight"); tp.beta = W32(c, "norm_cond.bias"); tp.cs = (const float2*)c->rope_cs.p;
  tp.tok_n = c->tokn.f(); tp.tok_r = c->tokr.f(); tp.B = B; tp.nseq = N; tp.d = d; tp.pos0 = c->pS0;
  if (d == 512) tpath_post_kernel<8><<<N + B, 256, 0, s>>>(tp);
  else tpath_post_kernel<4><<<N + B, 256, 0, s>>>(tp);
  // FiLM generators of every layer | K rows of the time tokens | V rows: independent of each other, one launch
  const SkinnyArgs g3[3] = {
      {c->mt.f(), d, c->film_w.f(), d, c->film_b.f(), c->film.f(), (int64_t)L * F * 2 * d, N, L * F * 2 * d, d, ACT_NONE},
      {c->tokr.f(), d, c->cak_w32.f(), d, c->cak_b.f(), c->ktail.f(), (int64_t)L * d, 2 * B, L * d, d, ACT_NONE},
      {c->tokn.f(), d, c->cav_w32.f(), d, c->cav_b.f(), c->vtail.f(), (int64_t)L * d, 2 * B, L * d, d, ACT_NONE}};
  CHK(launch_skinny3(g3, s));
  HIPCHK(hipGetLastError());
  return 0;
}

// dilated conv tail of the pose model (model/diffusion.py:214-224,398-402) as tap-accumulating GEMMs over
// left-padded per-sequence rows: buffers are [N][T+24][128|256] with the 24 pad rows zero.
static int pose_conv_tail(a2p_ctx* c, int N, int T, hipStream_t s) {
  const int C = c->C, hid = C > 256 ? C : 256, R = T + 24;
  const int M = N * R;
  const int ci[6] = {C, hid, C, C, C, C}, co[6] = {hid, C, C, C, C, C}, dil[6] = {1, 2, 3, 1, 2, 3};
  const int X = c->tail_x3 ? 3 : 1;   // split-operand rows are [hi | lo | hi]: three times as wide, contraction three times as long
  Buf* src = &c->cb[0];
  int cur = 0;
  for (int i = 0; i < 6; ++i) {
    const int cip = rup(ci[i], 64), cop = rup(co[i], 64);
    Buf* dst = (i == 0) ? &c->cb[1] : ((cur == 2) ? &c->cb[3] : &c->cb[2]);
    GemmP p = gemm_base(src->p, X * cip, c->conv_wt[i].p, X * cip, W32(c, "post_pose_layers." + std::to_string(i) + ".bias"), dst->p, X * cop, M,
                        co[i], X * cip);
    p.ntaps = 3;
    p.a_tap_stride = (int64_t)dil[i] * X * cip;
    p.w_tap_stride = (int64_t)co[i] * X * cip;
    p.epi = EPI_CONV;
    p.act = ACT_LRELU;
    if (c->tail_x3) p.split_third = cop;
    if (ci[i] == co[i]) {  // out = (in[t + 2*dil] + y) / 2
      p.skip = c->offT(*src, (int64_t)2 * dil[i] * X * cip);
      p.ld_skip = X * cip;
      if (c->tail_x3) p.skip_lo = cip;
    }
    CHK(launch_gemm(c, p, s));
    src = dst;
    cur = (dst == &c->cb[2]) ? 2 : (dst == &c->cb[3] ? 3 : 1);
  }
  GemmP pf = gemm_base(src->p, X * rup(C, 64), c->conv_wt[6].p, X * rup(C, 64), W32(c, "final_conv.bias"), c->mo.p, C, M, C, X * rup(C, 64));
  pf.out_f32 = 1;
  return launch_gemm(c, pf, s);
}

// The two launches of a forward that read CALLER memory (the timestep tensor, the noisy input): time embedding and input pack.
// They are the first launch of the time path and of the main path; forward_body(ext = false) leaves them out, so that what it
// launches touches context-owned buffers only and can be replayed as a captured graph (run_forward).
static int forward_ext(a2p_ctx* c, const float* x_in, const int64_t* t_orig, hipStream_t s) {
  const int d = c->d, B = c->pB, T = c->pT;
  time_embed_kernel<<<(B * (d / 2) + 255) / 256, 256, 0, s>>>(t_orig, c->time_freq.f(), c->emb.f(), B, d / 2);
  dim3 grid((T + 31) / 32, (c->Cpad + 31) / 32, B);
  if (c->tail_x3) pack_input_split3_kernel<<<grid, 256, 0, s>>>(x_in, (h16_t*)c->inpack.p, B, c->C, T, c->Cpad);
  else if (c->bf16 && !c->tail32) pack_input_kernel<h16_t><<<grid, 256, 0, s>>>(x_in, (h16_t*)c->inpack.p, B, c->C, T, c->Cpad);
  else pack_input_kernel<float><<<grid, 256, 0, s>>>(x_in, (float*)c->inpack.p, B, c->C, T, c->Cpad);
  HIPCHK(hipGetLastError());
  return 0;
}

// FiLMTransformer.forward for N sequences; result rows in c->mo ([N][mo_rows][C] fp32)
static int forward_body(a2p_ctx* c, const float* x_in, const int64_t* t_orig, int pass, int* mo_seq_rows, hipStream_t s, bool ext,
                        bool use_chain, bool use_small) {
  const int d = c->d, B = c->pB, T = c->pT, L = c->L, F = c->F;
  const int N = pass == A2P_PASS_CFG ? 2 * B : B;
  c->clk_turn = 0;
  const int* slots = (const int*)(pass == A2P_PASS_CFG ? c->slot_cfg.p : (pass == A2P_PASS_COND ? c->slot_cond.p : c->slot_unc.p));
  // The time path (7 latency-bound launches, ~70 us at B=8) is not needed before the first out_proj epilogue and can run on
  // the side stream next to input projection / norm1+QKV / self attention of layer 0 (-2..3 % step time).  On by default since
  // round 2 (A2P_NO_SIDE_STREAM=1 turns it off): in round 1, with the two queues active, 1-30 % of forwards on some boxes
  // differed for one sample; that was traced to tpath_post_kernel consuming a load right behind its s_waitcnt (kernels_misc.h,
  // docs/lab_notebook_r1_r4.md "Reproducibility") and fixed there -- 0 / 1500 differing forwards in round 1, 0 / 600 + identical 60-step
  // trajectories in both 16-bit modes in round 2 (scratch/side_stress.py).
  // (the small path keeps the time path on the main stream: on the side stream it measured 1430 against 1497 steps/s at 480 rows --
  // the fork / join costs more than the 7 launches it hides)
  const bool overlap_tpath = use_chain && !c->opt.no_side_stream;
  if (overlap_tpath) {
    c->ev_fork = c->ev_fork_pool[c->ev_turn & 7];
    c->ev_join = c->ev_join_pool[c->ev_turn & 7];
    ++c->ev_turn;
    HIPCHK(hipEventRecord(c->ev_fork, s));  // orders the side stream behind t_orig AND behind the previous step's readers
    HIPCHK(hipStreamWaitEvent(c->side, c->ev_fork, 0));
    CHK(time_path(c, t_orig, N, slots, c->side, ext));
    HIPCHK(hipEventRecord(c->ev_join, c->side));
    if (c->opt.side_early_join) HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));  // diagnostic: side stream without overlap
  } else {
    CHK(time_path(c, t_orig, N, slots, s, ext));
  }
  hipEvent_t tune0 = nullptr, tune1 = nullptr;
  if (use_chain) {   // (in front of the input projection: chain_in_wanted needs the workgroup shape; the calibration's timed region now includes the projection)
    c->ch_nw = chain_pick_nw(c, (int64_t)N * T, &tune0, &tune1);
    if (!tune0) chain_pick_family(c, (int64_t)N * T, &tune0, &tune1);   // (one calibration at a time)
    else c->ch_fam_mid = c->ch_fam_post = 1;
    if (tune0) HIPCHK(hipEventRecord(tune0, s));
  }
  // input permute + projection (model/diffusion.py:345-346,364); exact fp32 in every mode (a2p_ctx::tail32).
  // Chain forwards of the face model: inside the first chain kernel (chain4_kernel<MT, CHAIN_IN>: decoder_layer_chain), nothing to launch here.
  const bool fuse_in = use_chain && ext && chain_in_wanted(c, T, N == B || (N == 2 * B && !c->opt.no_shared_half));
  if (!fuse_in) {
    Fp32Scope f32(c, c->tail32 && !c->tail_x3);
    dim3 grid((T + 31) / 32, (c->Cpad + 31) / 32, B);
    const int X = c->tail_x3 ? 3 : 1;
    if (!ext) {   // forward_ext launched it
    } else if (c->tail_x3) pack_input_split3_kernel<<<grid, 256, 0, s>>>(x_in, (h16_t*)c->inpack.p, B, c->C, T, c->Cpad);
    else if (c->bf16) pack_input_kernel<h16_t><<<grid, 256, 0, s>>>(x_in, (h16_t*)c->inpack.p, B, c->C, T, c->Cpad);
    else pack_input_kernel<float><<<grid, 256, 0, s>>>(x_in, (float*)c->inpack.p, B, c->C, T, c->Cpad);
    const bool shared_half = use_chain && N == 2 * B && !c->opt.no_shared_half;
    GemmP p = gemm_base(c->inpack.p, X * c->Cpad, c->wt.at("input_projection.weight").p, X * c->Cpad, W32(c, "input_projection.bias"),
                        shared_half ? c->hff.p : c->x.p, d, B * T, d, X * c->Cpad);
    p.out_f32 = c->bf16 ? 1 : 0;   // gemm_kernel<float> stores fp32 either way; the flag only selects a 16-bit kernel instance
    // guidance without the shared layer-0 half (per-op and small-forward paths): the second half's copy of the rows is written by
    // the same epilogue (16-bit modes: the fp32-output store) instead of a copy kernel behind the GEMM
    const bool dup = N == 2 * B && !shared_half;
    const bool dup_in_epilogue = dup && c->bf16;
    if (dup_in_epilogue) p.dup_off = (int64_t)B * T * d;
    CHK(launch_gemm(c, p, s));
    if (dup && !dup_in_epilogue) {
      const int64_t n4 = (int64_t)B * T * d / 4;
      dup_rows_kernel<<<(int)((n4 + 255) / 256), 256, 0, s>>>(reinterpret_cast<const float4*>(c->x.p),
                                                             reinterpret_cast<float4*>(c->x.f() + (size_t)B * T * d), n4);
    }
  }
  CrossKV kv, kv2;
  bool fused_x3 = false;
  for (int l = 0; l < L; ++l) {
    kv.K = c->offT(c->kc, (int64_t)l * d); kv.k_slot_stride = (int64_t)c->Sld * L * d; kv.ldk = (int64_t)L * d;
    kv.VT = c->offT(c->vtc, (int64_t)l * d * c->Sld); kv.vt_slot_stride = (int64_t)L * d * c->Sld; kv.ldvt = c->Sld;
    kv.slots = slots; kv.S_main = c->pS0;
    kv.slot_rule = pass == A2P_PASS_CFG ? 3 : (pass == A2P_PASS_COND ? 1 : 2); kv.slot_b = B;   // slot_tables_kernel's contents
    kv.ktail = c->ktail.f() + (size_t)l * d; kv.vtail = c->vtail.f() + (size_t)l * d;
    kv.tail_row_stride = (int64_t)L * d; kv.tail_sample_stride = (int64_t)2 * L * d; kv.S_tail = 2; kv.tail_mod = B;
    if (c->pose) {
      kv2.K = c->offT(c->k2c, (int64_t)l * d); kv2.k_slot_stride = (int64_t)64 * L * d; kv2.ldk = (int64_t)L * d;
      kv2.VT = c->offT(c->vt2c, (int64_t)l * d * 64); kv2.vt_slot_stride = (int64_t)L * d * 64; kv2.ldvt = 64;
      kv2.slots = slots; kv2.S_main = c->pK; kv2.S_tail = 0; kv2.tail_mod = 1;
      kv2.slot_rule = kv.slot_rule; kv2.slot_b = B;
    }
    FilmRef fr;
    fr.base = c->film.f() + (size_t)l * F * 2 * d;
    fr.seq_stride = (int64_t)L * F * 2 * d;
    if (use_chain)
      CHK(decoder_layer_chain(c, l, N, T, kv, c->pose ? &kv2 : nullptr, fr, l == 0, l + 1 < L, s, (overlap_tpath && l == 0) ? c->ev_join : nullptr,
                              /*fuse_final=*/!c->pose && !c->tail32, /*shared_half=*/l == 0 && N == 2 * B && !c->opt.no_shared_half, &fused_x3, (l == 0 && fuse_in) ? x_in : nullptr));
    else if (use_small) CHK(decoder_layer_small(c, l, N, T, kv, fr, s, (overlap_tpath && l == 0) ? c->ev_join : nullptr));
    else CHK(decoder_layer(c, l, N, T, kv, c->pose ? &kv2 : nullptr, fr, s));
  }
  // (the family calibration times the decoder stack INCLUDING final_layer: the tall last-layer kernel may contain it)
  struct TuneEnd { hipEvent_t e; hipStream_t s; ~TuneEnd() { if (e) (void)hipEventRecord(e, s); } } tune_end{tune1, s};
  // final_layer (model/diffusion.py:397)   (A2P_TAIL16 + face + chain mode, or the tall last-layer kernel's split-operand island: already done by the last POST kernel)
  if (use_chain && !c->pose && (!c->tail32 || fused_x3)) {
    *mo_seq_rows = T;
    return 0;
  }
  if (c->pose && c->tail_fused) {   // final_layer + dilated conv stack + final_conv in ONE LDS-resident kernel (kernels_tail.h)
    TailP tp;
    memset(&tp, 0, sizeof(tp));
    tp.x = c->x.f(); tp.T = T; tp.d = d; tp.C = c->C; tp.nblk = (T + tail::TB - 1) / tail::TB;
    tp.w = reinterpret_cast<const h16_t*>(c->tail_w.p); tp.bias = c->tail_b.f(); tp.out = c->mo.f();
    for (int i = 0; i < tail::NLAYERS; ++i) tp.woff[i] = c->tail_woff[i];
    KernelTimer kt(c, A2P_KERNEL_GEMM, A2P_KERNEL_POSE_TAIL);
    A2P_LAUNCH(kt, pose_tail_kernel, N * tp.nblk, 512, s, tp);
    HIPCHK(hipGetLastError());
    *mo_seq_rows = T;
    return 0;
  }
  // fp32 GEMMs read the residual stream in place; the split-operand variant splits it, the all-16-bit variant casts it first
  Fp32Scope f32(c, c->tail32 && !c->tail_x3);
  const void* rows = c->x.p;
  const int X = c->tail_x3 ? 3 : 1;
  if (c->tail_x3) {
    const int64_t n = (int64_t)N * T * d;
    split3_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>(c->x.f(), d, 1, reinterpret_cast<h16_t*>(c->t3.p), (int64_t)N * T, d, d, 0);
    rows = c->t3.p;
  } else if (c->bf16) {
    CHK(launch_ln_rope(c, false, c->x.f(), d, nullptr, nullptr, c->xn.p, nullptr, d, N * T, T, 0, s));
    rows = c->xn.p;
  }
  if (!c->pose) {
    GemmP p = gemm_base(rows, X * d, c->wt.at("final_layer.weight").p, X * d, W32(c, "final_layer.bias"), c->mo.p, c->C, N * T, c->C, X * d);
    p.out_f32 = c->bf16 ? 1 : 0;
    CHK(launch_gemm(c, p, s));
    *mo_seq_rows = T;
  } else {
    GemmP p = gemm_base(rows, X * d, c->wt.at("final_layer.weight").p, X * d, W32(c, "final_layer.bias"), c->offT(c->cb[0], (int64_t)24 * X * 128),
                        X * 128, N * T, c->C, X * d);
    if (c->tail_x3) p.split_third = 128;
    p.rows_per_seq = T;
    p.out_seq_pad = 24;
    CHK(launch_gemm(c, p, s));
    CHK(pose_conv_tail(c, N, T, s));
    *mo_seq_rows = T + 24;
  }
  return 0;
}


// Opt-in (A2P_GRAPH=1): forwards that are not chain-kernel forwards (config 0: 76 launches of 4-12 us) replayed as a captured graph.
// Everything behind the two launches that read caller memory is captured ONCE per (pass, prepared geometry, kernel family) on a
// context-owned stream and replayed with hipGraphLaunch on the caller's stream; replays are bit-identical to stream launches
// (tests/test_hip_round3.py).  Measured (profiles/r03_ksplit_ab.txt): a dependent stream launch of an EMPTY kernel costs 2.8 us and is
// host-bound, a graph node 1.6 us (scratch/launch_floor.hip) -- but the step's kernels are longer than the host's enqueue time, the
// host runs ahead either way, and the step takes the same time (1742 vs 1757 steps/s); what the graph saves is ~200 us of host time per
// step.  Graphs die with the state they captured: a2p_finalize_weights / a2p_reload_env bump graph_epoch.  Never while a kernel class
// is being timed (dispatch-packet events are not capturable).
// Destroying an executable that a caller's stream may still be replaying is not something HIP promises to defer: a drop waits for the
// device first (rare: the 17th geometry of a long-lived context, a weight update, a2p_reload_env, a2p_ctx_destroy).
static void graphs_drop(a2p_ctx* c) {
  if (c->graphs.empty()) return;
  (void)hipDeviceSynchronize();
  for (auto& g : c->graphs) (void)hipGraphExecDestroy(g.exec);
  c->graphs.clear();
}

static int run_forward(a2p_ctx* c, const float* x_in, const int64_t* t_orig, int pass, int* mo_seq_rows, hipStream_t s) {
  if (!c->prepared) {
    set_err("denoise before a2p_prepare_cond");
    return A2P_ERR_STATE;
  }
  ARG(x_in && t_orig, "null argument");
  ARG(pass >= 0 && pass <= 2, "bad pass %d", pass);
  const int B = c->pB, T = c->pT;
  const int N = pass == A2P_PASS_CFG ? 2 * B : B;
  // Row panels pay off once there are enough of them: every workgroup streams the whole weight set of its chain, so a
  // forward of < ~1000 rows (config 0: B=1, T=240 -> 480 rows = 10 panels) is faster as many small 2-D tiles
  // (measured: 0.99 vs 1.10 ms per step at 480 rows, equal at 1200, chain ahead from 2400 rows on).
  // The kernel family is chosen from the row count of the UNSHARDED batch when the host names it (a2p_set_batch_hint;
  // sample_parallel does): the families differ in rounding, and a sharded run must reproduce the single-process samples bit for bit
  const int64_t rows_eff = (int64_t)(c->batch_hint > B ? (N / B) * c->batch_hint : N) * T;
  // (the 1100-row break-even was measured against the small-forward kernels, which exist for the face model only: a context
  // without them -- the body model, d != 512 -- keeps the round-2 threshold of 960 rows against the per-op kernels)
  const int64_t chain_thr = small_supported(c) ? c->opt.chain_rows : (c->opt.chain_rows < 960 ? c->opt.chain_rows : 960);
  const bool use_chain = chain_supported(c) && (rows_eff >= chain_thr || c->opt.chain_mt);
  const bool use_small = !use_chain && small_supported(c);
  if (use_chain || !c->opt.graph || c->time_kind >= 0) return forward_body(c, x_in, t_orig, pass, mo_seq_rows, s, true, use_chain, use_small);
  CHK(forward_ext(c, x_in, t_orig, s));
  const a2p_ctx::GraphKey key = {pass, B, T, c->pS0, c->pK, use_small ? 1 : 0, c->graph_epoch};
  for (auto& g : c->graphs)
    if (memcmp(&g.key, &key, sizeof(key)) == 0) {
      HIPCHK(hipGraphLaunch(g.exec, s));
      *mo_seq_rows = g.rows;
      return 0;
    }
  if (c->graphs.size() >= 16) graphs_drop(c);   // a long-lived context walking through many geometries
  if (!c->gstream) HIPCHK(hipStreamCreateWithFlags(&c->gstream, hipStreamNonBlocking));
  HIPCHK(hipStreamBeginCapture(c->gstream, hipStreamCaptureModeThreadLocal));
  int rows = 0;
  const int rc = forward_body(c, x_in, t_orig, pass, &rows, c->gstream, false, use_chain, use_small);
  hipGraph_t graph = nullptr;
  const hipError_t ce = hipStreamEndCapture(c->gstream, &graph);
  if (rc != 0 || ce != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    if (rc == 0) set_err("graph capture of the forward failed: %s", hipGetErrorString(ce));
    return rc != 0 ? rc : A2P_ERR_HIP;
  }
  hipGraphExec_t exec = nullptr;
  const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (ie != hipSuccess) {
    set_err("hipGraphInstantiate: %s", hipGetErrorString(ie));
    return A2P_ERR_HIP;
  }
  c->graphs.push_back({key, exec, rows});
  HIPCHK(hipGraphLaunch(exec, s));
  *mo_seq_rows = rows;
  return 0;
}

static int launch_step_tail(a2p_ctx* c, StepP& sp, hipStream_t s) {
  dim3 grid((sp.Tn + 31) / 32, (sp.C + 31) / 32, sp.B);
  step_tail_kernel<<<grid, 256, 0, s>>>(sp);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int a2p_denoise_forward(a2p_ctx* c, const float* x, const int64_t* t_orig, const float* scale, int32_t pass, float* out,
                                   void* stream) {
  ARG(c && out, "null argument");
  ARG(pass != A2P_PASS_CFG || scale, "A2P_PASS_CFG needs scale");
  hipStream_t s = (hipStream_t)stream;
  int rows = 0;
  CHK(run_forward(c, x, t_orig, pass, &rows, s));
  StepP sp;
  memset(&sp, 0, sizeof(sp));
  sp.mo = c->mo.f(); sp.mo_seq_rows = rows; sp.mo_ld = c->C; sp.B = c->pB; sp.C = c->C; sp.Tn = c->pT;
  sp.pass = pass; sp.scale = scale; sp.out_btc = out; sp.sampler = -1; sp.nonfinite = reinterpret_cast<int*>(c->nonfinite.p);
  return launch_step_tail(c, sp, s);
}

__global__ void map_timesteps_kernel(const int64_t* t_idx, const int64_t* tmap, int64_t* out, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) out[i] = tmap[t_idx[i]];
}

extern "C" int a2p_sample_step(a2p_ctx* c, int32_t sampler, const float* x, const int64_t* t_idx, const int64_t* timestep_map,
                               const float* tables, int32_t n_steps, const float* scale, const float* noise, float eta,
                               int32_t clip_denoised, float* x_next, float* pred_xstart, void* stream) {
  ARG(c && x && t_idx && timestep_map && tables && x_next, "null argument");
  ARG(sampler == A2P_SAMPLER_DDIM || sampler == A2P_SAMPLER_DDPM, "bad sampler");
  ARG(sampler == A2P_SAMPLER_DDIM || noise, "DDPM step needs noise");
  ARG(scale, "classifier-free guidance scale required");
  hipStream_t s = (hipStream_t)stream;
  // _WrappedModel.__call__ (respace.py:140-145): new_ts = timestep_map[ts]; reuse tmpa as int64 scratch
  int64_t* t_orig = reinterpret_cast<int64_t*>(c->tmpa.p);
  map_timesteps_kernel<<<1, 256, 0, s>>>(t_idx, timestep_map, t_orig, c->pB);
  int rows = 0;
  CHK(run_forward(c, x, t_orig, A2P_PASS_CFG, &rows, s));
  StepP sp;
  memset(&sp, 0, sizeof(sp));
  sp.mo = c->mo.f(); sp.mo_seq_rows = rows; sp.mo_ld = c->C; sp.B = c->pB; sp.C = c->C; sp.Tn = c->pT;
  sp.pass = A2P_PASS_CFG; sp.scale = scale; sp.sampler = sampler; sp.x = x; sp.t_idx = t_idx; sp.tables = tables;
  sp.n_steps = n_steps; sp.noise = noise; sp.eta = eta; sp.clip = clip_denoised; sp.x_next = x_next; sp.x0 = pred_xstart;
  sp.nonfinite = reinterpret_cast<int*>(c->nonfinite.p);
  return launch_step_tail(c, sp, s);
}

extern "C" int a2p_attention_logit_max(a2p_ctx* c, float* max_logit_host, void* stream) {
  ARG(c && max_logit_host, "null argument");
  hipStream_t s = (hipStream_t)stream;
  int v = 0;
  int* dev = reinterpret_cast<int*>(c->nonfinite.p) + 1;
  HIPCHK(hipMemcpyAsync(&v, dev, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipMemsetAsync(dev, 0x80, sizeof(int), s));
  if (v == (int)0x80808080) {   // no query-split attention launch since the last call
    *max_logit_host = -INFINITY;
    return 0;
  }
  const int bits = v >= 0 ? v : v ^ 0x7fffffff;
  memcpy(max_logit_host, &bits, sizeof(float));
  return 0;
}

extern "C" int a2p_precision_verdict(a2p_ctx* c, float* max_logit_host, int32_t* outside, void* stream) {
  ARG(c && max_logit_host && outside, "null argument");
  CHK(a2p_attention_logit_max(c, max_logit_host, stream));
  // fp32 contexts are exact at any magnitude; the 16-bit contexts are validated up to A2P_LOGIT_ENVELOPE_16BIT (a2p_hip.h)
  *outside = (c->bf16 && *max_logit_host > A2P_LOGIT_ENVELOPE_16BIT) ? 1 : 0;
  return 0;
}

extern "C" int a2p_check_finite(a2p_ctx* c, void* stream) {
  ARG(c, "null ctx");
  hipStream_t s = (hipStream_t)stream;
  int flag = 0;
  HIPCHK(hipMemcpyAsync(&flag, c->nonfinite.p, sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (!flag) return 0;
  HIPCHK(hipMemsetAsync(c->nonfinite.p, 0, sizeof(int), s));
  if (flag & 4) {
    set_err("a denoiser evaluation since the last check had a timestep outside [0, %d): the time-MLP table (a2p_finalize_weights) covers the reference's diffusion_steps; "
            "set A2P_TIME_TABLE=<rows> before loading the weights, or A2P_TIME_TABLE=0 to compute the time MLP in every forward", c->tct_rows);
    return A2P_ERR_ARG;
  }
  set_err("a denoiser evaluation since the last check produced inf / nan outputs (%s operands): the activations of this checkpoint "
          "leave the operand format's range, or the inputs / weights were not finite.  precision=\"bf16\" has fp32's range, "
          "precision=\"fp32\" is the parity mode",
#ifdef A2P_HALF
          c->bf16 ? "IEEE-half, |x| <= 65504" : "fp32"
#else
          c->bf16 ? "bfloat16" : "fp32"
#endif
  );
  return A2P_ERR_NONFINITE;
}

// ------------------------------------------------------------------------------------------------
// stand-alone sampler arithmetic
// ------------------------------------------------------------------------------------------------
extern "C" int a2p_p_mean_variance(const float* model_out, const float* x, const int64_t* t_idx, const float* tables, int32_t n_steps,
                                   int32_t batch, int32_t nfeats, int32_t frames, int32_t clip_denoised, float* pred_xstart,
                                   float* mean, void* stream) {
  ARG(model_out && x && t_idx && tables && pred_xstart, "null argument");
  StepP sp;
  memset(&sp, 0, sizeof(sp));
  sp.mo = model_out; sp.mo_seq_rows = frames; sp.mo_ld = nfeats; sp.B = batch; sp.C = nfeats; sp.Tn = frames;
  sp.pass = A2P_PASS_COND; sp.sampler = -1; sp.x = x; sp.t_idx = t_idx; sp.tables = tables; sp.n_steps = n_steps;
  sp.clip = clip_denoised; sp.x0 = pred_xstart; sp.mean = mean;
  return launch_step_tail(nullptr, sp, (hipStream_t)stream);
}

extern "C" int a2p_ddim_update(const float* pred_xstart, const float* x, const int64_t* t_idx, const float* tables, int32_t n_steps,
                               const float* noise, float eta, int32_t batch, int64_t per_sample, float* sample, void* stream) {
  ARG(pred_xstart && x && t_idx && tables && sample, "null argument");
  const int64_t total = batch * per_sample;
  ddim_update_kernel<<<(int)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(pred_xstart, x, t_idx, tables, n_steps, noise, eta,
                                                                                 per_sample, total, sample);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int a2p_p_sample_update(const float* mean, const int64_t* t_idx, const float* tables, int32_t n_steps, const float* noise,
                                   int32_t batch, int64_t per_sample, float* sample, void* stream) {
  ARG(mean && t_idx && tables && noise && sample, "null argument");
  const int64_t total = batch * per_sample;
  p_sample_update_kernel<<<(int)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(mean, t_idx, tables, n_steps, noise, per_sample,
                                                                                     total, sample);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int a2p_eps_from_xstart(const float* x, const float* pred_xstart, const int64_t* t_idx, const float* tables,
                                  int32_t n_steps, int32_t batch, int64_t per_sample, float* eps, void* stream) {
  ARG(x && pred_xstart && t_idx && tables && eps, "null argument");
  const int64_t total = batch * per_sample;
  eps_from_xstart_kernel<<<(int)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, pred_xstart, t_idx, tables, n_steps,
                                                                                     per_sample, total, eps);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int a2p_plms_update(const float* x, const float* pred_xstart, const int64_t* t_idx, const float* tables, int32_t n_steps,
                               const float* eps0, const float* eps1, const float* eps2, const float* eps3, int32_t mode,
                               int32_t batch, int64_t per_sample, float* sample, void* stream) {
  ARG(x && pred_xstart && t_idx && tables && eps0 && sample, "null argument");
  ARG(mode >= A2P_PLMS_PREDICT && mode <= A2P_PLMS_EULER, "bad PLMS mode %d", mode);
  const int need = mode == A2P_PLMS_EULER ? 2 : mode;  // eps buffers the mode reads beyond eps0
  ARG((need < 2 || eps1) && (need < 3 || eps2) && (need < 4 || eps3), "PLMS mode %d needs %d eps buffers", mode, need);
  const int64_t total = batch * per_sample;
  plms_update_kernel<<<(int)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, pred_xstart, t_idx, tables, n_steps, eps0, eps1,
                                                                                 eps2, eps3, mode, per_sample, total, sample);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int a2p_ddim_reverse_update(const float* pred_xstart, const float* x, const int64_t* t_idx, const float* tables,
                                       int32_t n_steps, int32_t batch, int64_t per_sample, float* sample, void* stream) {
  ARG(pred_xstart && x && t_idx && tables && sample, "null argument");
  const int64_t total = batch * per_sample;
  ddim_reverse_kernel<<<(int)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, pred_xstart, t_idx, tables, n_steps, per_sample,
                                                                                  total, sample);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int a2p_q_sample(const float* x_start, const int64_t* t_idx, const float* tables, int32_t n_steps, const float* noise,
                            int32_t batch, int64_t per_sample, float* out, void* stream) {
  ARG(x_start && t_idx && tables && noise && out, "null argument");
  const int64_t total = batch * per_sample;
  q_sample_kernel<<<(int)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x_start, t_idx, tables, n_steps, noise, per_sample, total,
                                                                              out);
  HIPCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------
// unit entry points
// ------------------------------------------------------------------------------------------------
extern "C" int a2p_gemm(a2p_ctx* c, const float* A, const float* W, const float* bias, float* C, int32_t M, int32_t N, int32_t K,
                        void* stream) {
  ARG(c && A && W && C && M > 0 && N > 0 && K > 0 && N % 4 == 0, "bad gemm arguments");
  hipStream_t s = (hipStream_t)stream;
  const int kp = rup(K, 64);
  Buf a, w;
  CHK(buf_alloc_tmp(a, (size_t)M * kp * c->esz));
  CHK(buf_alloc_tmp(w, (size_t)N * kp * c->esz));
  CHK(launch_cast(c, A, K, a.p, kp, M, K, kp, nullptr, s));
  CHK(launch_cast(c, W, K, w.p, kp, N, K, kp, nullptr, s));
  GemmP p = gemm_base(a.p, kp, w.p, kp, bias, C, N, M, N, kp);
  p.out_f32 = 1;
  int rc = launch_gemm(c, p, s);
  hipStreamSynchronize(s);
  buf_free(a);
  buf_free(w);
  return rc;
}

__global__ void widen_kernel(const h16_t* a, float* o, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = (float)a[i];
}

template <typename T>
__global__ void transpose_cast_kernel(const float* __restrict__ v, T* __restrict__ vt, int S, int d, int Sld) {
  // v [n][S][d] -> vt [n][d][Sld]
  const int n = blockIdx.z;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)S * d) return;
  const int s = (int)(i / d), c = (int)(i - (int64_t)s * d);
  vt[((int64_t)n * d + c) * Sld + s] = from_f32<T>(v[((int64_t)n * S + s) * d + c]);
}

extern "C" int a2p_attention(a2p_ctx* c, const float* q, const float* k, const float* v, float* out, int32_t N, int32_t Tq, int32_t S,
                             void* stream) {
  ARG(c && q && k && v && out && N > 0 && Tq > 0 && S > 0, "bad attention arguments");
  hipStream_t s = (hipStream_t)stream;
  const int d = c->d, Sld = rup(S, 64);
  Buf qb, kb, vb, ob;
  CHK(buf_alloc_tmp(qb, (size_t)N * Tq * d * c->esz));
  CHK(buf_alloc_tmp(kb, ((size_t)N * Sld + 64) * d * c->esz));
  CHK(buf_alloc_tmp(vb, (size_t)N * d * Sld * c->esz));
  CHK(buf_alloc_tmp(ob, (size_t)N * Tq * d * c->esz));
  CHK(launch_cast(c, q, d, qb.p, d, (int64_t)N * Tq, d, d, nullptr, s));
  for (int n = 0; n < N; ++n)
    CHK(launch_cast(c, k + (size_t)n * S * d, d, c->offT(kb, (int64_t)n * Sld * d), d, S, d, d, nullptr, s));
  dim3 tg((unsigned)(((int64_t)S * d + 255) / 256), 1, N);
  if (c->bf16) transpose_cast_kernel<h16_t><<<tg, 256, 0, s>>>(v, (h16_t*)vb.p, S, d, Sld);
  else transpose_cast_kernel<float><<<tg, 256, 0, s>>>(v, (float*)vb.p, S, d, Sld);
  AttnP a;
  memset(&a, 0, sizeof(a));
  a.Q = qb.p; a.q_seq_stride = (int64_t)Tq * d; a.ldq = d;
  a.K = kb.p; a.k_slot_stride = (int64_t)Sld * d; a.ldk = d;
  a.VT = vb.p; a.vt_slot_stride = (int64_t)d * Sld; a.ldvt = Sld;
  a.O = ob.p; a.o_seq_stride = (int64_t)Tq * d; a.ldo = d;
  a.tail_mod = 1; a.Tq = Tq; a.S_main = S; a.S_tail = 0;
  a.scale_log2e = 1.4426950408889634f / sqrtf((float)c->DH);
  int rc = launch_attn(c, a, N, A2P_KERNEL_ATTN_SELF, s);
  // back to fp32
  if (rc == 0) {
    if (c->bf16) {
      const int64_t n = (int64_t)N * Tq * d;
      widen_kernel<<<(int)((n + 255) / 256), 256, 0, s>>>((const h16_t*)ob.p, out, n);
    } else {
      hipMemcpyAsync(out, ob.p, (size_t)N * Tq * d * 4, hipMemcpyDeviceToDevice, s);
    }
  }
  hipStreamSynchronize(s);
  buf_free(qb); buf_free(kb); buf_free(vb); buf_free(ob);
  return rc;
}

extern "C" int a2p_decoder_layer_forward(a2p_ctx* c, int32_t layer, float* x, const float* memory, const float* t,
                                         const float* memory2, int32_t N, int32_t T, int32_t S, int32_t S2, void* stream) {
  ARG(c && x && memory && t, "null argument");
  if (!c->finalized) {
    set_err("a2p_decoder_layer_forward before a2p_finalize_weights");
    return A2P_ERR_STATE;
  }
  ARG(layer >= 0 && layer < c->L && N >= 1 && N <= c->Nmax && T >= 1 && T <= c->Tmax, "bad layer/shape");
  ARG(S >= 1 && S <= c->S0max + 2 && (int64_t)N * S <= c->rows_cap, "bad memory length");
  ARG(!memory2 || (c->pose && S2 >= 1 && S2 <= 64), "bad memory2");
  hipStream_t s = (hipStream_t)stream;
  const int d = c->d, F = c->F, Sld = rup(S, 64);
  const std::string p = "seqTransDecoder.stack." + std::to_string(layer) + ".";
  Buf kb, vb, k2b, v2b, mtb, flm;
  CHK(buf_alloc_tmp(kb, ((size_t)N * Sld + 64) * d * c->esz));
  CHK(buf_alloc_tmp(vb, (size_t)N * d * Sld * c->esz));
  CHK(buf_alloc_tmp(mtb, (size_t)N * d * 4));
  CHK(buf_alloc_tmp(flm, (size_t)N * F * 2 * d * 4));
  // K = rot(mem) Wk + bk ; V^T = (mem Wv + bv)^T   for this layer
  auto kvproj = [&](const float* mem, int len, int ld_rows, const std::string& an, Buf& kdst, Buf& vdst) -> int {
    CHK(launch_ln_rope(c, false, mem, d, nullptr, nullptr, c->xn.p, c->xr.p, d, N * len, len, 0, s));
    const Buf& inw = c->wt.at(p + an + ".in_proj_weight");
    const float* inb = W32(c, p + an + ".in_proj_bias");
    GemmP pk = gemm_base(c->xr.p, d, c->offT(inw, (int64_t)d * d), d, inb + d, kdst.p, d, N * len, d, d);
    pk.rows_per_seq = len;
    pk.out_seq_pad = ld_rows - len;
    CHK(launch_gemm(c, pk, s));
    GemmP pv = gemm_base(c->xn.p, d, c->offT(inw, (int64_t)2 * d * d), d, inb + 2 * d, vdst.p, ld_rows, N * len, d, d);
    pv.epi = EPI_STORE_T;
    pv.rows_per_seq = len;
    pv.t_seq_stride = (int64_t)d * ld_rows;
    return launch_gemm(c, pv, s);
  };
  CHK(kvproj(memory, S, Sld, "multihead_attn", kb, vb));
  CrossKV kv, kv2;
  kv.K = kb.p; kv.k_slot_stride = (int64_t)Sld * d; kv.ldk = d; kv.VT = vb.p; kv.vt_slot_stride = (int64_t)d * Sld; kv.ldvt = Sld;
  kv.S_main = S;
  if (memory2) {
    CHK(buf_alloc_tmp(k2b, ((size_t)N * 64 + 64) * d * c->esz));
    CHK(buf_alloc_tmp(v2b, (size_t)N * d * 64 * c->esz));
    CHK(kvproj(memory2, S2, 64, "multihead_attn2", k2b, v2b));
    kv2.K = k2b.p; kv2.k_slot_stride = (int64_t)64 * d; kv2.ldk = d; kv2.VT = v2b.p; kv2.vt_slot_stride = (int64_t)d * 64; kv2.ldvt = 64;
    kv2.S_main = S2;
  }
  // FiLM generators of this layer
  mish_kernel<<<(N * d + 255) / 256, 256, 0, s>>>(t, mtb.f(), (int64_t)N * d);
  CHK(launch_skinny(mtb.f(), d, c->film_w.f() + (size_t)layer * F * 2 * d * d, d, c->film_b.f() + (size_t)layer * F * 2 * d, flm.f(),
                    (int64_t)F * 2 * d, N, F * 2 * d, d, ACT_NONE, s));
  FilmRef fr;
  fr.base = flm.f();
  fr.seq_stride = (int64_t)F * 2 * d;
  HIPCHK(hipMemcpyAsync(c->x.p, x, (size_t)N * T * d * 4, hipMemcpyDeviceToDevice, s));
  int rc = chain_supported(c) ? decoder_layer_chain(c, layer, N, T, kv, memory2 ? &kv2 : nullptr, fr, true, false, s)
                              : decoder_layer(c, layer, N, T, kv, memory2 ? &kv2 : nullptr, fr, s);
  if (rc == 0) hipMemcpyAsync(x, c->x.p, (size_t)N * T * d * 4, hipMemcpyDeviceToDevice, s);
  hipStreamSynchronize(s);
  buf_free(kb); buf_free(vb); buf_free(k2b); buf_free(v2b); buf_free(mtb); buf_free(flm);
  return rc;
}

// ------------------------------------------------------------------------------------------------
// debugging aid (scratch/stress*.py): copy an internal buffer to the host after a device synchronise
// ------------------------------------------------------------------------------------------------
extern "C" int a2p_debug_read(a2p_ctx* c, const char* name, void* host, int64_t bytes) {
  ARG(c && name && host && bytes > 0, "bad arguments");
  const std::string n(name);
  if (n == "chain_nw") {   // int32: waves per chain workgroup of the last chain forward (4 | 8; bench.py reports it)
    ARG(bytes >= 4, "chain_nw is one int32");
    *reinterpret_cast<int32_t*>(host) = c->ch_nw;
    return 0;
  }
  if (n == "chain_family") {   // int32: chain kernel family of the last chain forward (10 x MID + POST; 1 = kernels_chain.h, 4 = kernels_chain4.h)
    ARG(bytes >= 4, "chain_family is one int32");
    *reinterpret_cast<int32_t*>(host) = c->ch_fam_mid * 10 + c->ch_fam_post;   // 11 | 44 | 41 | 14
    return 0;
  }
  if (n == "attn3_launches") {   // int64: launches of attn3_kernel (kernels_attn3.h) on this context so far (tests, bench)
    ARG(bytes >= 8, "attn3_launches is one int64");
    *reinterpret_cast<int64_t*>(host) = c->attn3_launches;
    return 0;
  }
  if (n == "chain_in_launches") {   // int64: launches of chain4_kernel<MT, CHAIN_IN> (input_projection + layer 0's PRE work) so far (tests)
    ARG(bytes >= 8, "chain_in_launches is one int64");
    *reinterpret_cast<int64_t*>(host) = c->in4_launches;
    return 0;
  }
  if (n == "final_fused_launches") {   // int64: last-layer tall POST kernels that computed final_layer as a split-operand island (tests)
    ARG(bytes >= 8, "final_fused_launches is one int64");
    *reinterpret_cast<int64_t*>(host) = c->fin_fused_launches;
    return 0;
  }
  if (n == "chain4_launches") {   // int64: launches of the tall chain kernels (kernels_chain4.h) on this context so far (tests, bench)
    ARG(bytes >= 8, "chain4_launches is one int64");
    *reinterpret_cast<int64_t*>(host) = c->ch4_launches;
    return 0;
  }
  const Buf* b = n == "film" ? &c->film : n == "ktail" ? &c->ktail : n == "vtail" ? &c->vtail : n == "tvec" ? &c->tvec
               : n == "tokr" ? &c->tokr : n == "tokn" ? &c->tokn : n == "tct" ? &c->tct
               : n == "x" ? &c->x : n == "qk" ? &c->qk : n == "vt" ? &c->vt : n == "ao" ? &c->ao : n == "mo" ? &c->mo
               : n == "clk" ? &c->clk : nullptr;
  ARG(b && (size_t)bytes <= b->bytes, "unknown buffer '%s' or too many bytes", name);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(host, b->p, (size_t)bytes, hipMemcpyDeviceToHost));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// measurement
// ------------------------------------------------------------------------------------------------
extern "C" int a2p_kernel_timing(a2p_ctx* c, int32_t kind, int32_t enable) {
  ARG(c, "null ctx");
  hipDeviceSynchronize();
  for (auto& e : c->evs) {
    hipEventDestroy(e.first);
    hipEventDestroy(e.second);
  }
  c->evs.clear();
  c->time_kind = enable ? kind : -1;
  return 0;
}

extern "C" int a2p_kernel_time_ms(a2p_ctx* c, double* total_ms, int64_t* launches) {
  ARG(c && total_ms && launches, "null argument");
  HIPCHK(hipDeviceSynchronize());
  double tot = 0;
  for (auto& e : c->evs) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e.first, e.second));
    tot += ms;
  }
  *total_ms = tot;
  *launches = (int64_t)c->evs.size();
  return 0;
}
