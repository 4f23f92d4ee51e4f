// Second attention kernel of the 16-bit modes (round 5): ONE 8-wave workgroup per CU, waves of unequal size, and a two-tile
// software pipeline inside every wave.  Same arithmetic per (query, key tile) as attn_kernel (kernels_attn.h: S^T = K Q^T,
// lane-local online softmax in the log2 domain, O^T += V^T P^T, the key <-> fragment map of AttnLds::krow), so the two kernels
// agree to the last bit for every query -- the difference is WHO computes WHAT WHEN:
//
//   * Balance.  attn_kernel gives a (sequence, head) pair of T = 600 queries five 4-wave workgroups of 128 queries; the
//     headline launch (16 sequences x 8 heads) is 640 workgroups = 2.5 per CU, i.e. half of the SIMDs carry three 32-query
//     waves and the other half two: the launch takes what 768 workgroups would (profiles/r04_attn_occupancy_probe.txt: 17 %
//     of the B=8 cross attention is imbalance).  Here a workgroup is 8 waves = 4 x 48 + 4 x 32 = 320 queries: two workgroups
//     per pair, 256 per launch, exactly one per CU, and every SIMD carries one 48-query and one 32-query wave (waves w and
//     w + 4 of a 512-thread workgroup share a SIMD): 5 query tiles per SIMD everywhere instead of 6 on the busy half.
//   * The dependency chain of a tile -- 16 QK^T MFMAs -> row maximum -> 32 exp -> convert -> 16 PV MFMAs -- is what the
//     3-waves-per-SIMD kernel leaves exposed (matrix pipe 20 % busy, VALU 40 %, profiles/r04_pmc_sq_counters.txt).  With two
//     256-register waves per SIMD a wave can hold TWO score tiles: S(t+1) = K(t+1) Q^T is issued BEFORE the softmax of S(t),
//     so the matrix pipe works under the wave's own VALU chain (cdna_hip_programming.md T15), and the PV MFMAs of query
//     tile q run under the softmax of query tile q + 1.
//   * One workgroup per CU leaves the LDS free for a 4-deep K / V^T tile ring: the DMA of tile t + 3 is issued while tile
//     t + 1 is consumed and waited for with a COUNTED vmcnt two steps later -- the per-tile barrier no longer doubles as a
//     DMA drain.
//
// Everything else (slot-indexed K/V, the time-token tail patched into the last tile(s), the XCD-aware grid, the non-temporal
// policy for per-sample K/V, the logit maximum, the output transposed through LDS and stored as whole rows) is attn_kernel's.
#pragma once
#include "kernels_attn.h"

template <int DH>
struct Attn2Geo {
  using L = AttnLds<h16_t, DH>;
  // NS ring slots: one workgroup per CU means the ring is the ONLY memory-level parallelism a CU has -- with 4 slots (2 tiles = 32 KB in
  // flight per CU) the kernel's skeleton (DMA, barriers, fragment reads; every MFMA and softmax instruction compiled out) alone took
  // 49.5 of the 75 us of the B=8 cross attention: latency-bound on the K/V stream (profiles/r05_attn2_ablation.txt).  8 slots at
  // DH = 64 (128 KB of the 160 KB LDS), 8 at DH = 32 (64 KB): 6 tiles in flight.
  static constexpr int NW = 8, NS = 8;                  // waves per workgroup, ring slots
  static constexpr int NPK = DH / 8, NPV = DH / 8;      // 1 KiB DMA pieces per K tile / V^T tile
  static constexpr int PW = (NPK + NPV) / NW;           // pieces per wave per tile (2 at DH = 64, 1 at DH = 32)
  static constexpr int SLOT = L::KSZ + L::VSZ;          // elements per ring slot (K tile, then V^T tile)
  static_assert((NPK + NPV) % NW == 0, "pieces must divide over the waves");
};

// one wave's share of the tile loop; QT = query tiles (of 16) this wave owns
template <int DH, int QT, int ABL>
__device__ __forceinline__ void attn2_wave(const AttnP& p, h16_t* const smem, const int q0, const int seq, const int head, const int slot,
                                           const int wid, const int lane, const int stage_row0, const int role) {
  using P = Prec<h16_t>;
  using G = Attn2Geo<DH>;
  using L = typename G::L;
  constexpr int KV = 64, NW = G::NW, NS = G::NS, PW = G::PW;
  constexpr int KC = DH / 32, DVT = DH / 16;
  const int l15 = lane & 15, g = lane >> 4;
  const bool kv_nt = p.kv_stream && slot != 0;
  const int S_total = p.S_main + p.S_tail;
  const int ntiles = (S_total + KV - 1) / KV;
  const bool wave_active = __builtin_amdgcn_readfirstlane(q0) < p.Tq;

  const h16_t* Qb = reinterpret_cast<const h16_t*>(p.Q) + (int64_t)seq * p.q_seq_stride + head * DH;
  const h16_t* Kb = reinterpret_cast<const h16_t*>(p.K) + (int64_t)slot * p.k_slot_stride + head * DH;
  const h16_t* Vb = reinterpret_cast<const h16_t*>(p.VT) + (int64_t)slot * p.vt_slot_stride + (int64_t)head * DH * p.ldvt;

  h16x8 qf[QT][KC];   // Q fragments, loaded in the prologue below

  // ---- tile DMA: piece pi of a tile = K piece pi (pi < NPK) or V^T piece pi - NPK; wave w owns pieces w, w + NW, ... ----
  constexpr int CPR = DH / 8, KRPI = 64 / CPR;   // 16-byte chunks per K row, K rows per piece
  int64_t src_off[PW];                           // element offset of this lane's 16 bytes inside the tile's K (or V^T) block
  int dst_off[PW];                               // wave-uniform LDS element offset inside a slot
  bool is_k[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int pi = wid + j * NW;
    is_k[j] = pi < G::NPK;
    if (is_k[j]) {
      const int kr = pi * KRPI + lane / CPR;
      src_off[j] = (int64_t)kr * p.ldk + (((lane % CPR) ^ L::kswz(kr)) << 3);
      dst_off[j] = pi * KRPI * DH;
    } else {
      const int vp = pi - G::NPK, vr = vp * 8 + (lane >> 3);
      src_off[j] = (int64_t)vr * p.ldvt + (((lane & 7) ^ L::vswz(vr)) << 3);
      dst_off[j] = L::KSZ + vp * 8 * KV;
    }
  }
  // The tile DMA is issued through INLINE ASM on purpose.  hipcc knows that __builtin_amdgcn_global_load_lds writes LDS and, for every
  // LDS read whose address it cannot prove disjoint from a pending DMA's destination (SIInsertWaitcnts: any run-time ring slot), puts
  // `s_waitcnt vmcnt(0)` in front of the read -- here that would be a wait for the tile requested a moment ago, in every step (seen in
  // the ISA of the first version of this kernel).  The hand-written counted vmcnt waits + barriers below are the ordering.
  // (M0 is written without being declared: hipcc rejects it as a clobber -- "reserved register" -- and nothing else in this kernel
  // uses M0: no LDS-DMA builtin, no readlane / movrel / sendmsg)
  auto dma16 = [&](const h16_t* src, h16_t* lds_wave_base, bool nt) __attribute__((always_inline)) {
    const uint32_t m0v = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) h16_t*)lds_wave_base;
    if (nt) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(m0v), "v"(src) : "memory");
    else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(src) : "memory");
  };
  auto issue_tile = [&](int tile, int slot) __attribute__((always_inline)) {
    h16_t* sl = smem + slot * G::SLOT;
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const h16_t* src = is_k[j] ? Kb + (int64_t)tile * KV * p.ldk + src_off[j] : Vb + tile * KV + src_off[j];
      dma16(src, sl + dst_off[j], kv_nt);
    }
  };
  // time-token rows of tile `tile` (block-uniform call site; every thread of the workgroup takes part)
  auto patch_tail = [&](int tile, int slot) __attribute__((always_inline)) {
    h16_t* Ks = smem + slot * G::SLOT;
    h16_t* Vs = Ks + L::KSZ;
    const int kv0 = tile * KV, sample = seq % p.tail_mod;
    for (int e = threadIdx.x; e < p.S_tail * DH; e += 64 * NW) {
      const int j = e / DH, c = e % DH;
      const int kl = p.S_main + j - kv0;
      if (kl >= 0 && kl < KV) {
        const int64_t off = (int64_t)sample * p.tail_sample_stride + (int64_t)j * p.tail_row_stride + head * DH + c;
        Ks[L::kidx(kl, c)] = (h16_t)p.ktail[off];
        Vs[L::vidx(c, kl)] = (h16_t)p.vtail[off];
      }
    }
  };
  auto has_tail = [&](int tile) __attribute__((always_inline)) { return p.S_tail > 0 && tile * KV + KV > p.S_main; };

  f32x4 o[QT][DVT], s[4][QT];
  h16x8 pf[QT][2];     // P fragments (keys 0..31 / 32..63 of the tile) of the wave's query tiles, between its softmax and its PV segment
  float mrun[QT], lsum[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    mrun[qt] = -INFINITY;
    lsum[qt] = 0.f;
#pragma unroll
    for (int dv = 0; dv < DVT; ++dv) o[qt][dv] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- the two segments a wave alternates between -------------------------------------------------------------------
  // MATRIX segment M(t): O^T += V^T(t) P(t)^T for all query tiles of the wave, then S^T = K(t+1) Q^T for the next tile -- MFMAs and
  // fragment reads only (each K / V^T fragment is read once and used for all QT query tiles).
  auto m_segment = [&](int t, auto has_next_c, auto masked_c) __attribute__((always_inline)) {
    constexpr bool HAS_NEXT = decltype(has_next_c)::value, MASKED = decltype(masked_c)::value;
    const h16_t* Vs = smem + (t & (NS - 1)) * G::SLOT + L::KSZ;
    // ALL fragment reads of the segment are issued up front, in MFMA order (hipcc, left alone, issues each ds_read_b128 two
    // instructions ahead of the MFMA that needs it: with only the partner's VALU segment beside it the wave then sits through an LDS
    // round trip per fragment -- the segment's skeleton, every MFMA compiled out, took as long as the kernel's: profiles/r05_attn2_ablation.txt)
    __builtin_amdgcn_sched_barrier(0);
    h16x8 vf[2][DVT];
    [[maybe_unused]] h16x8 kf[KC][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int dv = 0; dv < DVT; ++dv) {
        vf[c][dv] = *reinterpret_cast<const h16x8*>(&Vs[L::vidx(dv * 16 + l15, c * 32 + g * 8)]);
        if constexpr (MASKED) {   // never-written V^T padding columns: 0 x (inf | nan) = nan (kernels_attn.h load_vfr)
          const int nvalid = S_total - (t * KV + c * 32 + g * 8);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (e >= nvalid) vf[c][dv][e] = (h16_t)0.f;
        }
      }
    if constexpr (HAS_NEXT) {
      const h16_t* Ks = smem + ((t + 1) & (NS - 1)) * G::SLOT;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) kf[kc][kt] = P::load(&Ks[L::kidx(L::krow(kt, l15), kc * 32 + g * 8)]);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int dv = 0; dv < DVT; ++dv)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          if constexpr (!(ABL & 8)) o[qt][dv] = P::mfma(vf[c][dv], pf[qt][c], o[qt][dv]);
          else asm volatile("" : "+v"(o[qt][dv]) : "v"(vf[c][dv]), "v"(pf[qt][c]));
        }
    if constexpr (HAS_NEXT) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            if constexpr (!(ABL & 16)) s[kt][qt] = P::mfma(kf[kc][kt], qf[qt][kc], s[kt][qt]);
            else asm volatile("" : "+v"(s[kt][qt]) : "v"(kf[kc][kt]));
          }
    }
#ifndef ATTN2_NO_SGB
    if constexpr (!MASKED && !(ABL & 24)) {
      // issue order: the V^T fragment reads, then the PV MFMAs with the K fragment reads slotted between them, then the QK^T MFMAs
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * DVT, 0);
#pragma unroll
      for (int i = 0; i < 2 * DVT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, QT, 0);
        if (HAS_NEXT && i < 4 * KC) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      if constexpr (HAS_NEXT) __builtin_amdgcn_sched_group_barrier(0x008, 4 * KC * QT, 0);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
  };
  // VECTOR segment V(t): the online softmax of the wave's scores of tile t (log2 domain): s -> P fragments, running max / sum,
  // O rescaled -- VALU only.
  auto v_segment = [&](int t, auto masked_c) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_c)::value;
    if constexpr (MASKED) {
      const int kv0 = t * KV;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kv0 + L::krow(kt, g * 4 + r) >= S_total) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) s[kt][qt][r] = -INFINITY;
          }
    }
#pragma unroll
    for (int q = 0; q < QT; ++q) {
      float mx = s[0][q][0];
      if constexpr (!(ABL & 2)) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][q][r]);
        mx = attn_rowgroup_max(mx);
      }
      const float mnew = fmaxf(mrun[q], mx * p.scale_log2e);
      const float alpha = __builtin_amdgcn_exp2f(mrun[q] - mnew);
      mrun[q] = mnew;
      f32x2 ps2 = {0.f, 0.f};
      const f32x2 sc2 = {p.scale_log2e, p.scale_log2e}, mn2 = {-mnew, -mnew};
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x2 v = {s[kt][q][2 * h], s[kt][q][2 * h + 1]};
          if constexpr (!(ABL & 32)) v = __builtin_elementwise_fma(v, sc2, mn2);
          if constexpr (!(ABL & 1)) {
            v[0] = __builtin_amdgcn_exp2f(v[0]);
            v[1] = __builtin_amdgcn_exp2f(v[1]);
          }
          if constexpr (!(ABL & 32)) ps2 += v;
          s[kt][q][2 * h] = v[0];
          s[kt][q][2 * h + 1] = v[1];
        }
      lsum[q] = lsum[q] * alpha + (ps2[0] + ps2[1]);
      if constexpr (!(ABL & 32)) {
#pragma unroll
        for (int dv = 0; dv < DVT; ++dv)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[q][dv][r] *= alpha;
      }
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pf[q][c][r] = (h16_t)s[2 * c][q][r];
          pf[q][c][4 + r] = (h16_t)s[2 * c + 1][q][r];
        }
    }
  };

  // ---- prologue: tile 0 and Q landed, tiles 1 .. NS-2 requested, S(0) computed by every wave ----
  issue_tile(0, 0);
  // Q fragments (B operand of S^T = K Q^T): lane (query l15, k-group g).  Loaded AFTER the first tile requests and then touched: the
  // compiler's own scoreboard must see these loads retired before the tile loop -- it cannot see the hand-written DMA and vmcnt waits,
  // and left to itself it carries "six loads pending" into the loop and emits vmcnt(5) .. vmcnt(0) in front of the first uses of qf in
  // EVERY iteration, which the hardware reads as "drain the tile DMA" (seen in the ISA).
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    int q = q0 + qt * 16 + l15;
    if (q >= p.Tq) q = p.Tq - 1;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) qf[qt][kc] = P::load(Qb + (int64_t)q * p.ldq + kc * 32 + g * 8);
  }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) asm volatile("" : "+v"(qf[qt][kc]));   // (hipcc puts its s_waitcnt vmcnt(0) here: Q and tile 0 landed)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int t = 1; t < NS - 1; ++t)
    if (t < ntiles) issue_tile(t, t);
  __builtin_amdgcn_s_barrier();
  if (has_tail(0)) {
    patch_tail(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const bool last_partial = ntiles * KV > S_total;
  // S(0): the M segment "before tile 0" (no PV yet: P = 0 fragments would do, but the PV MFMAs are simply skipped)
  if (wave_active) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      pf[qt][0] = h16x8{};
      pf[qt][1] = h16x8{};
    }
    const h16_t* Ks = smem;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) s[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const h16x8 kf = P::load(&Ks[L::kidx(L::krow(kt, l15), kc * 32 + g * 8)]);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s[kt][qt] = P::mfma(kf, qf[qt][kc], s[kt][qt]);
      }
  }

  // ---- phases ----------------------------------------------------------------------------------------------------------
  // The two waves of a SIMD (w and w + 4: role 0 and role 1) alternate between the VECTOR and the MATRIX segment IN ANTI-PHASE,
  // one workgroup barrier per phase:
  //     phase 2t     role 0: V(t)          role 1: M(t-1)
  //     phase 2t+1   role 0: M(t)          role 1: V(t)
  // so that on every SIMD one wave streams MFMAs while the other runs the softmax: an in-order wave that wants the matrix pipe
  // while its partner holds it stalls WITH its VALU work behind it (SQ_WAIT_INST_ANY 28-35 % in both earlier forms of the kernel,
  // profiles/r05_attn2_v1_pmc.txt) -- segments of one kind per wave and a partner of the other kind remove that collision
  // (MI355X_MICROARCH.md "Two waves per SIMD"; cdna_hip_programming.md T16).
  // Ring bookkeeping happens at the ODD phase starts: phase 2t+1 needs tile t+1 landed (K(t+1) is read by M(t)) and may refill
  // the slot of tile t-1 (with tile t+NS-1), whose last reader was role 1's M(t-1) in phase 2t.
  // (ABL 64: no tile DMA / DMA waits after the prologue, ABL 128: no phase barriers -- timing experiments only, results are wrong)
  auto sync_even = [&]() __attribute__((always_inline)) { if constexpr (!(ABL & 128)) __builtin_amdgcn_s_barrier(); };
  auto sync_odd = [&](int t) __attribute__((always_inline)) {
    if constexpr ((ABL & 64) != 0) {
      if constexpr (!(ABL & 128)) __builtin_amdgcn_s_barrier();
      return;
    }
    // tile t+1 landed for this wave; the tiles requested after it (t+2 .. t+NS-2, as far as they exist) may stay in flight
    {
      const int younger = min(ntiles - 2 - t, NS - 3);   // wave-uniform
      if (younger >= NS - 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * (NS - 3)) : "memory");
      else if (younger == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * 4) : "memory");
      else if (younger == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * 3) : "memory");
      else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * 2) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    static_assert(NS - 3 <= 5 && NS >= 4, "the wait ladder covers up to 5 younger tiles");
    __builtin_amdgcn_s_barrier();
    if (t + NS - 1 < ntiles) issue_tile(t + NS - 1, (t + NS - 1) & (NS - 1));
    if (t + 1 < ntiles && has_tail(t + 1)) {
      patch_tail(t + 1, (t + 1) & (NS - 1));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  };
  const std::false_type F{};
  const std::true_type Tt{};
  auto do_v = [&](int t) __attribute__((always_inline)) {
    if (!wave_active) return;
    if (t + 1 == ntiles && last_partial) v_segment(t, Tt);
    else v_segment(t, F);
  };
  auto do_m = [&](int t) __attribute__((always_inline)) {
    if (!wave_active) return;
    if (t + 1 < ntiles) m_segment(t, Tt, F);
    else if (last_partial) m_segment(t, F, Tt);
    else m_segment(t, F, F);
  };
  if (role == 0) {
    for (int t = 0; t < ntiles; ++t) {
      sync_even();
      do_v(t);
      sync_odd(t);
      do_m(t);
    }
    sync_even();                     // phase 2 * ntiles: role 1's last matrix segment
  } else {
    for (int t = 0; t < ntiles; ++t) {
      sync_even();
      if (t > 0) do_m(t - 1);
      sync_odd(t);
      do_v(t);
    }
    sync_even();
    do_m(ntiles - 1);
  }

  if (p.stat_max && wave_active) {
    float m = mrun[0];
#pragma unroll
    for (int qt = 1; qt < QT; ++qt) m = fmaxf(m, mrun[qt]);
#pragma unroll
    for (int sh = 8; sh > 0; sh >>= 1) m = fmaxf(m, __shfl_xor(m, sh, 64));
    const int mi = attn_ordered_int(m * 0.6931471805599453f);
    if (lane == 0 && mi > __atomic_load_n(p.stat_max, __ATOMIC_RELAXED)) atomicMax(p.stat_max, mi);
  }

  // ---- normalise, transpose through the (idle) ring, store whole rows ----
  constexpr int SP = DH + 8, PPR = DH / 8, NPC = QT * 16 * PPR / 64;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // every wave is done reading the last tile
  h16_t* stw = smem + stage_row0 * SP;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float l = lsum[qt];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
#pragma unroll
    for (int dv = 0; dv < DVT; ++dv) {
      const f32x4 v = o[qt][dv];
      *reinterpret_cast<h16x4*>(stw + (qt * 16 + l15) * SP + dv * 16 + g * 4) =
          h16x4{(h16_t)(v[0] * inv), (h16_t)(v[1] * inv), (h16_t)(v[2] * inv), (h16_t)(v[3] * inv)};
    }
  }
  h16_t* Ob = reinterpret_cast<h16_t*>(p.O) + (int64_t)seq * p.o_seq_stride + head * DH;
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int pc = lane + 64 * i, row = pc / PPR, part = pc % PPR;
    const h16x8 v = *reinterpret_cast<const h16x8*>(stw + row * SP + part * 8);
    const int q = q0 + row;
    if (q < p.Tq) *reinterpret_cast<h16x8*>(Ob + (int64_t)q * p.ldo + part * 8) = v;
  }
}

// grid: p.nq * p.nheads * p.nseq workgroups of 512 threads, p.nq = ceil(Tq / (4 * 16 * (QTA + QTB)))
// PAIR: which two waves are taken to share a SIMD (they get opposite roles): 0 = waves w and w + 4 (a workgroup's waves go to the SIMDs
// in cyclic order, MI355X_MICROARCH.md), 1 = waves 2k and 2k + 1 (scratch/attn2_bench measures both)
template <int DH, int QTA = 3, int QTB = 2, int ABL = 0, int PAIR = 0>
__global__ __launch_bounds__(512, 2) void attn2_kernel(AttnP p) {
  using G = Attn2Geo<DH>;
  constexpr int BQ = 4 * 16 * (QTA + QTB);
  __shared__ __attribute__((aligned(16))) h16_t smem[G::NS * G::SLOT];
  static_assert(BQ * (DH + 8) <= G::NS * G::SLOT, "output staging does not fit the tile ring");
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int qb, head, seq;
  {
    const int b = blockIdx.x;
    if (p.xcd_remap) {
      const int xcd = b & 7, j = b >> 3, pair = (j / p.nq) * 8 + xcd;
      qb = j % p.nq;
      head = pair % p.nheads;
      seq = pair / p.nheads;
    } else {
      qb = b % p.nq;
      head = (b / p.nq) % p.nheads;
      seq = b / (p.nq * p.nheads);
    }
  }
  const int slot = attn_slot(p, seq);
  const int role = PAIR == 0 ? (wid >> 2) : (wid & 1), idx = PAIR == 0 ? (wid & 3) : (wid >> 1);
  if (role == 0) {
    const int r0 = idx * (16 * QTA);
    attn2_wave<DH, QTA, ABL>(p, smem, qb * BQ + r0, seq, head, slot, wid, lane, r0, 0);
  } else {
    const int r0 = 4 * 16 * QTA + idx * (16 * QTB);
    attn2_wave<DH, QTB, ABL>(p, smem, qb * BQ + r0, seq, head, slot, wid, lane, r0, 1);
  }
}
