// Third attention kernel of the 16-bit modes (round 6): ONE wave per SIMD, 80 queries per wave, and a tile body written at ISA level.
//
// What rounds 2-5 measured about attn_kernel / attn2_kernel (docs/lab_notebook_r1_r4.md 4.2, docs/lab_notebook_r5.md 4): per 64-key
// tile a 32-query wave issues 32 MFMAs against ~220 VALU instructions (6.8 per MFMA), the loop is instruction-ISSUE bound, and every
// term of the ablation adds.  Round 6 measured the issue model itself (scratch/ubench/mfma_fill.hip, profiles/r06_mfma_fill.txt): in
// one wave's in-order stream a v_mfma_f32_16x16x32 (18 cycles back to back) hides TWO single-issue instructions; every further one
// costs 4 cycles, a v_exp_f32 counts as two.  So the levers are (1) fewer vector instructions per score and (2) exactly the right
// instructions between the MFMAs -- which a compiler-scheduled kernel cannot be talked into (see `step`).
//
//   1. VALU instructions per score.  attn_kernel pays, per score: scale-and-subtract fma, exp2, row-sum add, max, convert, AND one
//      multiply of the O accumulator (head_dim 64 = 64 O rows per query against 64 keys per tile: the unconditional `o *= alpha` is as
//      many instructions as the exponentials).  Here
//        * Q is multiplied by log2(e)/sqrt(dh) ONCE when its fragments are loaded;
//        * the running reference m_ref of the online softmax enters the score tile through the MFMA's C operand (S = K Q^T - m_ref:
//          the accumulator is initialised with -m_ref instead of 0), so P = exp2(S) needs no subtract;
//        * m_ref is LAZY (guide T13): it only moves when some score exceeds it by more than THR = 8 (P <= 256: exact in the fp32 sums,
//          the same relative rounding in the 16-bit P operand); the test is one wave vote per tile step and the rare fix-up (rescale O
//          and l, recompute or shift the pending score tile) is its own block OUTSIDE the fast loop.  Softmax is invariant to the
//          reference value: the result differs from the exact-max form by rounding only.
//        * the ROW SUMS are matrix work (v5): l[q] += ones x P[q]^T, one extra MFMA per query tile and 32-key chunk whose A operand is a constant
//          fragment of 16-bit ones, instead of 16 v_add_f32 per query tile and key tile -- the step is vector-issue bound (without its MFMAs it
//          takes MORE cycles than with them, profiles/r06_attn3_step_ablation_v4.txt) and the matrix pipe is half idle;
//        * keys past the end of a partial LAST tile are masked in that ones fragment, not in the scores (v6): their K rows are copies of the
//          tile's key 0, their V^T columns zero, so the last tile is an ordinary step and nothing is recomputed for it.
//      Fast path per score: exp2, half a max3, half a convert = 2 instructions (3 issue slots) instead of ~6.8.
//   2. The stream.  For the five query tiles q of a wave, "group q" is 18 MFMAs -- O^T[q] += V^T(t) P[q]^T (8) and l[q] += ones x P[q]^T (2), then
//      S[q] = K(t+1) Q[q]^T - m_ref[q] (8, overwriting the consumed scores in place) -- with the softmax of query tile q+1 and the maxima
//      of the S[q-1] written one group earlier issued between them, packet by packet.  Per step: 90 MFMAs (18 cycles each, 2 hidden issue
//      slots each) against 80 exp2 + 40 max3 + 40 cvt_pk + 16 fragment reads = 256 slots: 90 x 18 + 76 x 4 = 1924 cycles by the model, 2498
//      measured with the ring (tile DMA issue, barrier) and loop control (docs/lab_notebook_r6.md sections 3, 4, 7).
//
// Geometry: workgroup = 4 waves x 5 query tiles of 16 = 320 queries; T = 600 -> 2 workgroups per (sequence, head) pair; the headline
// launch (16 sequences x 8 heads) is 256 workgroups = exactly one per CU, every SIMD carries 5 query tiles (attn_kernel: 6 on the
// busy half).  K / V^T tiles of 64 keys arrive by LDS-DMA into an 8-slot ring, inline asm + counted vmcnt.  Fragment layouts,
// swizzles, the key <-> fragment map, slot-indexed K/V, time-token tail, XCD-aware grid, non-temporal policy, logit maximum and the
// LDS-transposed store are attn_kernel's.
//
// REGISTER OWNERSHIP.  The score tile (80 registers), the P fragments, the five row-sum accumulator tiles and the five running maxima live in
// v[A3_OWN : 255], O, Q, the K / V^T fragment sets and the ones fragments in the accumulator file; all are addressed LITERALLY by the asm statements; hipcc is held below A3_OWN (amdgpu_num_vgpr) and never sees
// them.  History of why (each seen in the ISA of a build of this file): builtins + sched_barrier(0) after every packet came out
// with all 80 exponentials of a step in front of its first MFMA (instruction selection linearises pure values before the scheduler
// ever sees the barriers); asm volatile statements keep their order, but with the tile as a C++ value hipcc parked the loop-carried
// tile in the accumulator file around the loop header (80 v_accvgpr_write + 80 v_accvgpr_read per step) whenever ANY other code in
// the kernel -- a second step variant, the fix-up block, the partial last tile -- touched it, and copied fresh MFMA results two
// instructions behind their (to it, opaque) producer, i.e. before they had landed.  Owned registers end all of that.
#pragma once
#include "kernels_attn.h"

template <int B, int E, class F>
__device__ __forceinline__ void attn3_static_for(F&& f) {   // f(integral_constant<int, B>) ... f(integral_constant<int, E - 1>)
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    attn3_static_for<B + 1, E>(f);
  }
}

template <int DH>
struct Attn3Geo {
  using L = AttnLds<h16_t, DH>;
  static constexpr int NW = 4, NS = 8;                  // waves per workgroup, ring slots
  static constexpr int NPK = DH / 8, NPV = DH / 8;      // 1 KiB DMA pieces per K tile / V^T tile
  static constexpr int PW = (NPK + NPV) / NW;           // pieces per wave per tile (4 at DH = 64, 2 at DH = 32)
  static constexpr int SLOT = L::KSZ + L::VSZ;          // elements per ring slot (K tile, then V^T tile)
  static_assert((NPK + NPV) % NW == 0, "pieces must divide over the waves");
};

#define A3_QT 5
// Owned registers.  Vector file, v[A3_OWN : 255] (hipcc allocates v0 .. v[A3_OWN - 1]):
#define A3_OWN 100
#define A3_PF(b, c) (100 + 4 * (2 * (b) + (c)))   // P fragment tuples (MFMA B operand), three buffers: b = q % 3, c = key chunk (query tile 0 of the NEXT tile
                                                  // is converted while query tile 4's fragments, buffer 1, are still being read)
#define A3_CI(q) (124 + 4 * (q))                  // -m_ref of query tile q, four copies: the C operand of its first QK^T MFMA
#define A3_AL(q) (144 + 4 * (q))                  // row sums of query tile q as an MFMA accumulator tile: l[query l15] in every register of every lane group
#define A3_M(q) (168 + (q))                       // running maximum of query tile q's scores RELATIVE to its reference (per lane)
#define A3_SB 176                                 // score (kt, q, r): v[A3_SB + 4 * (kt * A3_QT + q) + r]; tuples are MFMA C/D operands
#define A3_S(kt, q, r) (A3_SB + 4 * ((kt) * A3_QT + (q)) + (r))
// Accumulator file, a[0 : 247] (hipcc is given no reason to touch the accumulator file at all: every "a" value is literal):
#define A3_AO(q, dv, DVT_) (4 * ((q) * (DVT_) + (dv)))          // O^T accumulators a[0 : 79]
#define A3_AQ(q, kc, KC_) (80 + 4 * ((q) * (KC_) + (kc)))       // Q fragments a[80 : 119]
// two fragment sets (a[120 : 183], a[184 : 247]): step t consumes set t & 1 while the fragments of step t + 1 land in the other one
#define A3_AK(set, kc, kt) (120 + 64 * (set) + 4 * ((kc) * 4 + (kt)))           // K fragments (of the tile whose scores the step produces)
#define A3_AV(set, c, dv, DVT_) (152 + 64 * (set) + 4 * ((c) * (DVT_) + (dv)))  // V^T fragments (of the tile the step consumes)
#define A3_ONES(c) (248 + 4 * (c))   // A operand of the row-sum MFMAs of 32-key chunk c: ones -- for the LAST tile: ones on its valid keys, zeros on the padding
#ifndef A3X
#define A3X 0   // scratch timing experiments (results wrong): 1 no exp2, 2 no maxima, 4 no softmax packets at all, 8 no MFMAs, 16 no fragment reads
#endif
#ifndef A2P_ATTN3_THR
#define A2P_ATTN3_THR 8.0f     // log2 units a score may exceed the lazy reference by before the reference moves
#endif
#ifdef A2P_HALF
#define A3_MFMA "v_mfma_f32_16x16x32_f16"
#define A3_CVT "v_cvt_pk_f16_f32"
#else
#define A3_MFMA "v_mfma_f32_16x16x32_bf16"
#define A3_CVT "v_cvt_pk_bf16_f32"
#endif
// wait states behind an asm MFMA before anything but an accumulating MFMA may touch its result (nobody pads inside or behind asm:
// cdna_hip_programming.md 5.7; 8-pass MFMA -> VALU)
#define A3_MFMA_LANDED() asm volatile("s_nop 15\n\ts_nop 7" ::: "memory")

// ABL (scratch/attn3_bench.hip only): 64 no tile DMA after the prologue, 128 no barriers
template <int DH, int ABL = 0>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(A3_OWN))) void attn3_kernel(AttnP p) {
  using P = Prec<h16_t>;
  using G = Attn3Geo<DH>;
  using L = typename G::L;
  constexpr int QT = A3_QT, KV = 64, NW = G::NW, NS = G::NS, PW = G::PW, BQ = NW * QT * 16;
  constexpr int KC = DH / 32, DVT = DH / 16;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) h16_t smem[NS * G::SLOT];
  static_assert(BQ * (DH + 8) <= NS * G::SLOT, "output staging does not fit the tile ring");
  asm volatile("" ::: "v255", "a255");   // (the kernel descriptor must cover the owned registers of both files)

  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, g = lane >> 4;
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
  const int q0 = qb * BQ + wid * (QT * 16);
  const bool kv_nt = p.kv_stream && slot != 0;
  const int S_total = p.S_main + p.S_tail;
  const int ntiles = (S_total + KV - 1) / KV;
  const int rem = S_total % KV;                   // keys of a partial last tile (0: the last tile is full)
  const bool wave_active = __builtin_amdgcn_readfirstlane(q0) < p.Tq;

  const h16_t* Qb = reinterpret_cast<const h16_t*>(p.Q) + (int64_t)seq * p.q_seq_stride + head * DH;
  const h16_t* Kb = reinterpret_cast<const h16_t*>(p.K) + (int64_t)slot * p.k_slot_stride + head * DH;
  const h16_t* Vb = reinterpret_cast<const h16_t*>(p.VT) + (int64_t)slot * p.vt_slot_stride + (int64_t)head * DH * p.ldvt;

  // ---- tile DMA (attn2_kernel's: inline asm so that hipcc neither waits for it nor orders LDS reads behind it) ----
  // piece pi of a tile = K piece pi (pi < NPK) or V^T piece pi - NPK; wave w owns pieces w, w + NW, ...: its first NPK / NW are K pieces
  constexpr int CPR = DH / 8, KRPI = 64 / CPR, KPW = G::NPK / NW;
  int64_t src_off[PW];
  int dst_off[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int pi = wid + j * NW;
    if (j < KPW) {
      const int kr = pi * KRPI + lane / CPR;
      src_off[j] = (int64_t)kr * p.ldk + (((lane % CPR) ^ L::kswz(kr)) << 3);
      dst_off[j] = pi * KRPI * DH;
    } else {
      const int vp = pi - G::NPK, vr = vp * 8 + (lane >> 3);
      src_off[j] = (int64_t)vr * p.ldvt + (((lane & 7) ^ L::vswz(vr)) << 3);
      dst_off[j] = L::KSZ + vp * 8 * KV;
    }
  }
  // tiles are requested strictly in order: the (wave-uniform) source bases advance by one tile per request, the per-lane part is a
  // 32-bit byte offset (global_load_lds with an SGPR base: two scalar adds per tile instead of four 64-bit vector adds)
  unsigned dma_off[PW];
#pragma unroll
  for (int j = 0; j < PW; ++j) dma_off[j] = (unsigned)(src_off[j] * 2);
  const char* knext = reinterpret_cast<const char*>(Kb);
  const char* vnext = reinterpret_cast<const char*>(Vb);
  const int64_t kstep = (int64_t)KV * p.ldk * 2;
  auto issue_tile = [&](int rs) __attribute__((always_inline)) {
    h16_t* sl = smem + rs * G::SLOT;
    if (kv_nt) {   // block-uniform: ONE branch per tile
#pragma unroll
      for (int j = 0; j < PW; ++j) {
        const uint32_t m0v = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) h16_t*)(sl + dst_off[j]);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(m0v), "v"(dma_off[j]), "s"(j < KPW ? knext : vnext) : "memory");
      }
    } else {
#pragma unroll
      for (int j = 0; j < PW; ++j) {
        const uint32_t m0v = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) h16_t*)(sl + dst_off[j]);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(dma_off[j]), "s"(j < KPW ? knext : vnext) : "memory");
      }
    }
    knext += kstep;
    vnext += KV * 2;
  };
  // the last tile(s) in LDS, by the whole workgroup (block-uniform call sites): the time-token rows are written into it (attn_kernel),
  // and the rows / columns of keys past the end are made harmless -- K rows become copies of the tile's key 0 (their scores are
  // real, finite duplicates: they cannot raise a running maximum; the scores themselves are masked to -inf before the exponentials),
  // V^T columns become 0 (they are never-written memory: 0 x (inf | nan) = nan, kernels_attn.h load_vfr)
  auto finish_last_tile = [&](int tile, int rs) __attribute__((always_inline)) {
    h16_t* Ks = smem + rs * G::SLOT;
    h16_t* Vs = Ks + L::KSZ;
    const int kv0 = tile * KV;
    if (p.S_tail > 0 && kv0 + KV > p.S_main) {
      const int sample = seq % p.tail_mod;
      for (int e = threadIdx.x; e < p.S_tail * DH; e += 64 * NW) {
        const int j = e / DH, c = e % DH;
        const int kl = p.S_main + j - kv0;
        if (kl >= 0 && kl < KV) {
          const int64_t off = (int64_t)sample * p.tail_sample_stride + (int64_t)j * p.tail_row_stride + head * DH + c;
          Ks[L::kidx(kl, c)] = (h16_t)p.ktail[off];
          Vs[L::vidx(c, kl)] = (h16_t)p.vtail[off];
        }
      }
    }
    const int nvalid = S_total - kv0;
    if (nvalid < KV) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // (key 0 may itself be a time token written above)
      for (int e = threadIdx.x; e < (KV - nvalid) * DH; e += 64 * NW) {
        const int kl = nvalid + e / DH, c = e % DH;
        Ks[L::kidx(kl, c)] = Ks[L::kidx(0, c)];
        Vs[L::vidx(c, kl)] = (h16_t)0.f;
      }
    }
  };
  // does anything have to be written into `tile` once it has landed?  (time tokens: the last tile, or the last two when S_main % 64 == 63)
  auto needs_finish = [&](int tile) __attribute__((always_inline)) {
    return tile < ntiles && ((p.S_tail > 0 && tile * KV + KV > p.S_main) || (tile + 1 == ntiles && rem > 0));
  };

#ifdef A3_STAMPS   // scratch: where does a step's time go?  s_memtime deltas summed over the steps of workgroup 0 / wave 0 (p.kv_slot = output)
  long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long stamp_last = 0;
  const long long stamp_t0 = __builtin_readcyclecounter();
#define A3_STAMP(k) do { const long long now_ = __builtin_readcyclecounter(); stamp_acc[k] += now_ - stamp_last; stamp_last = now_; } while (0)
#else
#define A3_STAMP(k) do { } while (0)
#endif
#ifdef A3_WGSTAMPS   // scratch: per-WORKGROUP timeline (wave 0): 100 MHz wall clock at kernel start / loop entry / loop end / kernel end + the loop's shader cycles
  long long wg_t[6] = {(long long)wall_clock64(), 0, 0, 0, 0, 0};
#define A3_WGSTAMP(k) do { wg_t[k] = (long long)wall_clock64(); } while (0)
#define A3_WGCYC(k) do { wg_t[k] = __builtin_readcyclecounter(); } while (0)
#else
#define A3_WGSTAMP(k) do { } while (0)
#define A3_WGCYC(k) do { } while (0)
#endif
  // first tile that has something written into it after it landed (time tokens, padding of the partial tile)
  int tfin = ntiles;
  if (rem > 0) tfin = ntiles - 1;
  if (p.S_tail > 0) tfin = min(tfin, p.S_main / KV);

  // ---- state ----
  float mref[QT];            // (the only per-query state hipcc sees)
#pragma unroll
  for (int q = 0; q < QT; ++q) mref[q] = 0.f;
  attn3_static_for<0, QT>([&](auto q_c) __attribute__((always_inline)) {
    constexpr int q = decltype(q_c)::value;
    // l = 0, running maximum = -inf, reference 0
    asm volatile("v_mov_b32 v[%c0], 0xff800000" ::"n"(A3_M(q)));
    asm volatile("v_mov_b32 v[%c0], 0\n\tv_mov_b32 v[%c1], 0\n\tv_mov_b32 v[%c2], 0\n\tv_mov_b32 v[%c3], 0" ::"n"(A3_AL(q)), "n"(A3_AL(q) + 1), "n"(A3_AL(q) + 2), "n"(A3_AL(q) + 3));
    asm volatile("v_mov_b32 v[%c0], 0\n\tv_mov_b32 v[%c1], 0\n\tv_mov_b32 v[%c2], 0\n\tv_mov_b32 v[%c3], 0" ::"n"(A3_CI(q)), "n"(A3_CI(q) + 1), "n"(A3_CI(q) + 2), "n"(A3_CI(q) + 3));
    attn3_static_for<0, 4 * DVT>([&](auto e_c) __attribute__((always_inline)) {
      constexpr int e = decltype(e_c)::value;
      asm volatile("v_accvgpr_write_b32 a[%c0], 0" ::"n"(A3_AO(q, 0, DVT) + e));
    });
  });

#ifdef A2P_HALF
  constexpr unsigned ONE1 = 0x3c00u;   // an IEEE-half one
#else
  constexpr unsigned ONE1 = 0x3f80u;   // a bfloat16 one
#endif
  attn3_static_for<0, 8>([&](auto e_c) __attribute__((always_inline)) {
    asm volatile("v_accvgpr_write_b32 a[%c0], %1" ::"n"(A3_ONES(0) + decltype(e_c)::value), "v"(ONE1 | (ONE1 << 16)));
  });
  // LDS byte addresses of this lane's K / V^T fragments inside ring slot 0.  The swizzle of a fragment depends on the lane only, the
  // k-chunk (kc / c) flips one bit of the swizzled chunk index (hence one base per chunk), kt / dv are plain row offsets (immediates)
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) h16_t*)smem;
  unsigned kbase[KC], vbase[2];
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) kbase[kc] = smem_base + 2u * (unsigned)L::kidx(L::krow(0, l15), kc * 32 + g * 8);
#pragma unroll
  for (int c = 0; c < 2; ++c) vbase[c] = smem_base + 2u * (unsigned)(L::KSZ + L::vidx(l15, c * 32 + g * 8));

  // K / V^T fragments of the tile in ring slot rs -> their accumulator registers (ds_read_b128 straight into the accumulator file:
  // MFMA A operands may live there)
  auto read_kf = [&](unsigned rs, auto set_c) __attribute__((always_inline)) {
    attn3_static_for<0, KC>([&](auto k_c) __attribute__((always_inline)) {
      constexpr int kc = decltype(k_c)::value;
      (void)&kbase;
      const unsigned a = kbase[kc] + rs * (unsigned)(G::SLOT * 2);
      attn3_static_for<0, 4>([&](auto t_c) __attribute__((always_inline)) {
        constexpr int kt = decltype(t_c)::value, R = A3_AK(decltype(set_c)::value, kc, kt);
        (void)&a;
        if constexpr (!(A3X & 16)) asm volatile("ds_read_b128 a[%c0:%c1], %2 offset:%3" ::"n"(R), "n"(R + 3), "v"(a), "n"((32 * (kt >> 1) + 4 * (kt & 1)) * L::LSK * 2));
      });
    });
  };
  auto read_vf = [&](unsigned rs, auto set_c) __attribute__((always_inline)) {
    attn3_static_for<0, 2>([&](auto c_c) __attribute__((always_inline)) {
      constexpr int c = decltype(c_c)::value;
      (void)&vbase;
      const unsigned a = vbase[c] + rs * (unsigned)(G::SLOT * 2);
      attn3_static_for<0, DVT>([&](auto d_c) __attribute__((always_inline)) {
        constexpr int dv = decltype(d_c)::value, R = A3_AV(decltype(set_c)::value, c, dv, DVT);
        (void)&a;
        if constexpr (!(A3X & 16)) asm volatile("ds_read_b128 a[%c0:%c1], %2 offset:%3" ::"n"(R), "n"(R + 3), "v"(a), "n"(dv * 16 * L::LSV * 2));
      });
    });
  };
  auto wait_lds = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

  // ---- cold code on the owned registers (prologue, reference moves, the partial last tile): plain sequences, nothing interleaved ----
  // S = K(tile in slot rs) Q^T - m_ref for all query tiles
  auto qk_owned = [&](unsigned rs) __attribute__((always_inline)) {   // (uses fragment set 0: every call is followed by a `prime`)
    read_kf(rs, std::integral_constant<int, 0>{});
    wait_lds();
    attn3_static_for<0, KC>([&](auto k_c) __attribute__((always_inline)) {
      attn3_static_for<0, 4>([&](auto t_c) __attribute__((always_inline)) {
        attn3_static_for<0, QT>([&](auto q_c) __attribute__((always_inline)) {
          constexpr int kc = decltype(k_c)::value, kt = decltype(t_c)::value, q = decltype(q_c)::value, R = A3_S(kt, q, 0);
          if constexpr (kc == 0) asm volatile(A3_MFMA " v[%c0:%c1], a[%c2:%c3], a[%c4:%c5], v[%c6:%c7]" ::"n"(R), "n"(R + 3), "n"(A3_AK(0, kc, kt)), "n"(A3_AK(0, kc, kt) + 3), "n"(A3_AQ(q, kc, KC)), "n"(A3_AQ(q, kc, KC) + 3), "n"(A3_CI(q)), "n"(A3_CI(q) + 3));
          else asm volatile(A3_MFMA " v[%c0:%c1], a[%c2:%c3], a[%c4:%c5], v[%c0:%c1]" ::"n"(R), "n"(R + 3), "n"(A3_AK(0, kc, kt)), "n"(A3_AK(0, kc, kt) + 3), "n"(A3_AQ(q, kc, KC)), "n"(A3_AQ(q, kc, KC) + 3));
        });
      });
    });
    A3_MFMA_LANDED();
  };
  // keys past the end of the LAST tile need no score mask: their K rows are copies of the tile's key 0 (finish_last_tile: finite duplicate scores that cannot
  // raise a maximum), their V^T columns are zero (no contribution to O), and the ones fragment of that tile's row-sum MFMAs is zero on them (no
  // contribution to l) -- so the partial tile is an ordinary step and nothing is recomputed for it
  auto mask_last_ones = [&]() __attribute__((always_inline)) {
    const int kv0 = (ntiles - 1) * KV;
    A3_MFMA_LANDED();   // (the previous step's row-sum MFMAs have long read the fragment; cheap insurance in a cold path)
    attn3_static_for<0, 8>([&](auto e_c) __attribute__((always_inline)) {
      constexpr int e = decltype(e_c)::value, c = e >> 2, w = e & 3, kt = 2 * c + (w >> 1), r0 = (w & 1) * 2;   // P fragment register w of chunk c = scores (kt, r0), (kt, r0 + 1)
      const unsigned m = (kv0 + L::krow(kt, g * 4 + r0) < S_total ? ONE1 : 0u) | (kv0 + L::krow(kt, g * 4 + r0 + 1) < S_total ? ONE1 << 16 : 0u);
      asm volatile("v_accvgpr_write_b32 a[%c0], %1" ::"n"(A3_ONES(0) + e), "v"(m));
    });
    asm volatile("s_nop 7" ::: "memory");   // (v_accvgpr_write -> MFMA operand)
  };
  auto max_owned = [&]() __attribute__((always_inline)) {   // fold the tile into the running maxima
    attn3_static_for<0, QT>([&](auto q_c) __attribute__((always_inline)) {
      attn3_static_for<0, 8>([&](auto j_c) __attribute__((always_inline)) {
        constexpr int q = decltype(q_c)::value, j = decltype(j_c)::value, R = A3_S(j >> 1, q, (j & 1) * 2);
        asm volatile("v_max3_f32 v[%c0], v[%c0], v[%c1], v[%c2]" ::"n"(A3_M(q)), "n"(R), "n"(R + 1));
      });
    });
  };
  // does any running maximum of the wave exceed its reference by more than THR?
  auto must_move = [&]() __attribute__((always_inline)) {
    float a, b;
    asm volatile("v_max3_f32 %0, v[%c1], v[%c2], v[%c3]" : "=v"(a) : "n"(A3_M(0)), "n"(A3_M(1)), "n"(A3_M(2)));
    asm volatile("v_max3_f32 %0, %1, v[%c2], v[%c3]" : "=v"(b) : "v"(a), "n"(A3_M(3)), "n"(A3_M(4)));
    return __any(b > A2P_ATTN3_THR);
  };
  // The references move to the exact running row maxima: O and l rescaled, the running maxima re-based; the pending score tile is
  // shifted in place (SHIFT) or left to the caller to recompute from the K tile that is still in the ring.  Rare after the first tile.
  auto move_refs = [&](auto shift_c, auto first_c) __attribute__((always_inline)) {
    constexpr bool SHIFT = decltype(shift_c)::value, FIRST = decltype(first_c)::value;   // FIRST: O and l are still zero
    A3_MFMA_LANDED();
    attn3_static_for<0, QT>([&](auto q_c) __attribute__((always_inline)) {
      constexpr int q = decltype(q_c)::value;
      (void)&mref;
      float m;
      asm volatile("v_mov_b32 %0, v[%c1]" : "=v"(m) : "n"(A3_M(q)));
      const float d = attn_rowgroup_max(m);    // over the 4 lanes that share the query column
      const float f = __builtin_amdgcn_exp2f(-d);
      mref[q] += d;
      const float nm = -mref[q];
      asm volatile("v_sub_f32 v[%c0], v[%c0], %1" ::"n"(A3_M(q)), "v"(d));
      if constexpr (!FIRST)
        asm volatile("v_mul_f32 v[%c0], v[%c0], %4\n\tv_mul_f32 v[%c1], v[%c1], %4\n\tv_mul_f32 v[%c2], v[%c2], %4\n\tv_mul_f32 v[%c3], v[%c3], %4"
                     ::"n"(A3_AL(q)), "n"(A3_AL(q) + 1), "n"(A3_AL(q) + 2), "n"(A3_AL(q) + 3), "v"(f));
      asm volatile("v_mov_b32 v[%c0], %4\n\tv_mov_b32 v[%c1], %4\n\tv_mov_b32 v[%c2], %4\n\tv_mov_b32 v[%c3], %4" ::"n"(A3_CI(q)), "n"(A3_CI(q) + 1), "n"(A3_CI(q) + 2), "n"(A3_CI(q) + 3), "v"(nm));
      attn3_static_for<0, FIRST ? 0 : 4 * DVT>([&](auto e_c) __attribute__((always_inline)) {   // O *= f (through a vector register: the accumulator file has no arithmetic)
        constexpr int e = decltype(e_c)::value;
        (void)&f;
        float x;
        asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(x) : "n"(A3_AO(q, 0, DVT) + e));
        x *= f;
        asm volatile("v_accvgpr_write_b32 a[%c0], %1" ::"n"(A3_AO(q, 0, DVT) + e), "v"(x));
      });
      if constexpr (SHIFT) {
        attn3_static_for<0, 16>([&](auto n_c) __attribute__((always_inline)) {
          constexpr int n = decltype(n_c)::value;
          (void)&d;
          asm volatile("v_sub_f32 v[%c0], v[%c0], %1" ::"n"(A3_S(n >> 2, q, n & 3)), "v"(d));
        });
      }
    });
    asm volatile("s_nop 7" ::: "memory");   // (v_accvgpr_write / VALU writes -> MFMA operands: wait states nobody else inserts)
  };

  // ---- prologue ----
  issue_tile(0);
  {
    h16x8 qf[QT][KC];          // Q fragments (B operand of S^T = K Q^T): lane (query l15, k-group g), pre-multiplied by log2(e) / sqrt(dh)
#pragma unroll
    for (int q = 0; q < QT; ++q) {
      int qi = q0 + q * 16 + l15;
      if (qi >= p.Tq) qi = p.Tq - 1;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) qf[q][kc] = P::load(Qb + (int64_t)qi * p.ldq + kc * 32 + g * 8);
    }
    attn3_static_for<0, QT>([&](auto q_c) __attribute__((always_inline)) {
      attn3_static_for<0, KC>([&](auto k_c) __attribute__((always_inline)) {
        constexpr int q = decltype(q_c)::value, kc = decltype(k_c)::value;
        (void)&qf;
        asm volatile("" : "+v"(qf[q][kc]));   // hipcc's own vmcnt(0) for the Q loads lands here, not inside the loop (attn2_kernel)
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[q][kc][e] = (h16_t)((float)qf[q][kc][e] * p.scale_log2e);
        const u32x4 w = __builtin_bit_cast(u32x4, qf[q][kc]);
        asm volatile("v_accvgpr_write_b32 a[%c0], %4\n\tv_accvgpr_write_b32 a[%c1], %5\n\tv_accvgpr_write_b32 a[%c2], %6\n\tv_accvgpr_write_b32 a[%c3], %7"
                     ::"n"(A3_AQ(q, kc, KC)), "n"(A3_AQ(q, kc, KC) + 1), "n"(A3_AQ(q, kc, KC) + 2), "n"(A3_AQ(q, kc, KC) + 3), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]));
      });
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int t = 1; t < NS - 1; ++t)
    if (t < ntiles) issue_tile(t);
  __builtin_amdgcn_s_barrier();
  if (needs_finish(0)) {
    finish_last_tile(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- one tile step ----------------------------------------------------------------------------------------------------------------
  // Query-tile-major software pipeline, written out in ISSUE order (an in-order wave overlaps vector and matrix work only if they
  // alternate in its instruction stream).  Every instruction is `asm volatile`: volatile statements keep their SOURCE ORDER.  hipcc
  // still allocates the registers of the values it can see ("a" = accumulator file: O, Q and the K / V^T fragments, which ds_read_b128
  // loads straight into it; "v": P fragments, -m_ref) and pads nothing: every reader of an MFMA result sits at least one MFMA group
  // (>= 8 MFMAs) behind its producer, or behind A3_MFMA_LANDED.
  // QK = false: the drain behind the loop (softmax + O^T += V^T P^T of the last tile only).
  // softmax packet n of query tile q, in issue order: exp2 of score n, the row-sum add of score n - 2, the convert of score pair
  // (n - 5) / 2 into the P fragment registers of the query tile.  Every index is a compile-time constant (attn3_static_for).
  auto soft = [&](auto n_c, auto q_c) __attribute__((always_inline)) {
    constexpr int n = decltype(n_c)::value, q = decltype(q_c)::value;
    if constexpr (n < 16 && !(A3X & 5)) asm volatile("v_exp_f32 v[%c0], v[%c0]" ::"n"(A3_S(n >> 2, q, n & 3)));
    if constexpr (n >= 5 && ((n - 5) & 1) == 0 && (n - 5) / 2 < 8 && !(A3X & 4)) {
      constexpr int j = (n - 5) / 2, c = j >> 2, w = j & 3, kt = 2 * c + (w >> 1), r0 = (w & 1) * 2;
      asm volatile(A3_CVT " v[%c0], v[%c1], v[%c2]" ::"n"(A3_PF(q % 3, c) + w), "n"(A3_S(kt, q, r0)), "n"(A3_S(kt, q, r0 + 1)));
    }
  };
  constexpr int NSOFT = 21;   // n = 0 .. 20 covers 16 exps, 16 adds, 8 converts
  // the softmax of query tile 0, on its own (loop entry / re-entry and the drain; inside the loop it runs EARLY, see `step`)
  auto soft0_now = [&]() __attribute__((always_inline)) {
    attn3_static_for<0, NSOFT>([&](auto n_c) __attribute__((always_inline)) { soft(n_c, std::integral_constant<int, 0>{}); });
  };
  // step t consumes fragment set PAR = t & 1 (V^T(t), K(next tile): requested one step earlier, or by `prime`) and requests the
  // fragments of step t + 1 into the other set, one ds_read_b128 per MFMA gap of query tile 0's group (sixteen of them back to back
  // from four lockstep waves block each wave's issue for ~40 cycles apiece).
  // QK = true (the loop): the softmax of query tile 0 of tile t is ALREADY DONE (by the previous step, or soft0_now), and the one of
  // tile t + 1 is issued between the MFMAs of this step's LAST group, whose MFMA order is QK^T first, PV second, so that the maxima of
  // its new scores fit behind them: nothing of a step is left un-overlapped (measured before: softmax of query tile 0 347 of 3000
  // cycles per step, wait states + tail maxima ~70).  The early softmax runs BEFORE the vote on tile t + 1: if that tile turns out to
  // need a reference move it is recomputed from the ring anyway (the row sums live on the matrix pipe: the early softmax adds nothing to l).
  // QK = false: the drain of the LAST tile (PV + row-sum MFMAs with the softmax of query tiles 1 .. 4 between them; query tile 0 is the caller's).
  auto step = [&](int t, auto qk_c, auto par_c) __attribute__((always_inline)) {
    constexpr bool QK = decltype(qk_c)::value;
    constexpr int PAR = decltype(par_c)::value;
    A3_STAMP(6);  // (loop control + must_move since the last stamp)
    wait_lds();   // the fragments of THIS step (requested a step ago: long landed)
    [[maybe_unused]] unsigned fa_v[2], fa_k[KC];
    if constexpr (QK) {
      const unsigned vs = (unsigned)(min(t + 1, ntiles - 1) & (NS - 1)) * (unsigned)(G::SLOT * 2);
      const unsigned ks = (unsigned)(min(t + 2, ntiles - 1) & (NS - 1)) * (unsigned)(G::SLOT * 2);
#pragma unroll
      for (int c = 0; c < 2; ++c) fa_v[c] = vbase[c] + vs;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) fa_k[kc] = kbase[kc] + ks;
    }
    auto frag_read = [&](auto i_c) __attribute__((always_inline)) {   // read number i of the next step's fragments
      constexpr int i = decltype(i_c)::value;
      (void)&fa_v; (void)&fa_k;
      if constexpr (A3X & 16) return;
      if constexpr (i < 2 * DVT) {
        constexpr int c = i / DVT, dv = i % DVT, R = A3_AV(PAR ^ 1, c, dv, DVT);
        asm volatile("ds_read_b128 a[%c0:%c1], %2 offset:%3" ::"n"(R), "n"(R + 3), "v"(fa_v[c]), "n"(dv * 16 * L::LSV * 2));
      } else {
        constexpr int kc = (i - 2 * DVT) / 4, kt = (i - 2 * DVT) % 4, R = A3_AK(PAR ^ 1, kc, kt);
        asm volatile("ds_read_b128 a[%c0:%c1], %2 offset:%3" ::"n"(R), "n"(R + 3), "v"(fa_k[kc]), "n"((32 * (kt >> 1) + 4 * (kt & 1)) * L::LSK * 2));
      }
    };
    A3_STAMP(0);  // fragment wait + addresses
    constexpr int NSUM = 2;   // row-sum MFMAs per group: l[q] += ones x P[q]^T, one per 32-key chunk
    constexpr int SPC = DVT + NSUM / 2;          // PV-part MFMAs per 32-key chunk
    constexpr int NPV = 2 * SPC, NQK = QK ? 4 * KC : 0, NM = NPV + NQK;   // MFMAs per group
    attn3_static_for<0, QT>([&](auto q_c) __attribute__((always_inline)) {
      constexpr int q = decltype(q_c)::value;
      constexpr int qp = q > 0 ? q - 1 : 0;
      constexpr bool LASTG = QK && q == QT - 1;   // the last group: QK^T MFMAs first
      attn3_static_for<0, NM>([&](auto i_c) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value;
        constexpr bool is_pv = LASTG ? i >= NQK : i < NPV;
        constexpr int ip = LASTG ? i - NQK : i, iq = LASTG ? i : i - NPV;
        // PV part, per 32-key chunk c: the DVT MFMAs of O^T[q] and the row-sum MFMA behind them -- the two accumulating MFMAs of
        // an accumulator are DVT + 1 apart, never back to back
        if constexpr (is_pv && ip % SPC == DVT) {
          constexpr int c = ip / SPC, RL = A3_AL(q);
          if constexpr (!(A3X & 8)) asm volatile(A3_MFMA " v[%c0:%c1], a[%c2:%c3], v[%c4:%c5], v[%c0:%c1]" ::"n"(RL), "n"(RL + 3), "n"(A3_ONES(c)), "n"(A3_ONES(c) + 3), "n"(A3_PF(q % 3, c)), "n"(A3_PF(q % 3, c) + 3));
        } else if constexpr (is_pv) {
          constexpr int c = ip / SPC, dv = ip % SPC, RO = A3_AO(q, dv, DVT);
          if constexpr (!(A3X & 8)) asm volatile(A3_MFMA " a[%c0:%c1], a[%c2:%c3], v[%c4:%c5], a[%c0:%c1]" ::"n"(RO), "n"(RO + 3), "n"(A3_AV(PAR, c, dv, DVT)), "n"(A3_AV(PAR, c, dv, DVT) + 3), "n"(A3_PF(q % 3, c)), "n"(A3_PF(q % 3, c) + 3));
        } else {
          constexpr int kc = iq / 4, kt = iq % 4, R = A3_S(kt, q, 0);
          if constexpr (A3X & 8) {}
          else if constexpr (kc == 0) asm volatile(A3_MFMA " v[%c0:%c1], a[%c2:%c3], a[%c4:%c5], v[%c6:%c7]" ::"n"(R), "n"(R + 3), "n"(A3_AK(PAR, kc, kt)), "n"(A3_AK(PAR, kc, kt) + 3), "n"(A3_AQ(q, kc, KC)), "n"(A3_AQ(q, kc, KC) + 3), "n"(A3_CI(q)), "n"(A3_CI(q) + 3));
          else asm volatile(A3_MFMA " v[%c0:%c1], a[%c2:%c3], a[%c4:%c5], v[%c0:%c1]" ::"n"(R), "n"(R + 3), "n"(A3_AK(PAR, kc, kt)), "n"(A3_AK(PAR, kc, kt) + 3), "n"(A3_AQ(q, kc, KC)), "n"(A3_AQ(q, kc, KC) + 3));
        }
        // fillers: the softmax of query tile q + 1 (behind the last group: of query tile 0 of the NEXT tile) spread over the group's
        // gaps, the maxima of the S[q - 1] written one group earlier in every other gap, in query tile 0's group one fragment read of
        // the next step per gap, and behind the last group's QK^T MFMAs the maxima of its own new scores
        if constexpr (QK && q == 0 && i < 2 * DVT + 4 * KC) frag_read(i_c);
        if constexpr (q + 1 < QT) {
          constexpr int n0 = (i * NSOFT) / NM, n1 = ((i + 1) * NSOFT) / NM;
          attn3_static_for<n0, n1>([&](auto n_c) __attribute__((always_inline)) { soft(n_c, std::integral_constant<int, q + 1>{}); });
        } else if constexpr (QK) {
          constexpr int n0 = (i * NSOFT) / NM, n1 = ((i + 1) * NSOFT) / NM;
          attn3_static_for<n0, n1>([&](auto n_c) __attribute__((always_inline)) { soft(n_c, std::integral_constant<int, 0>{}); });
        }
        if constexpr (QK && q > 0 && !(A3X & 2)) {
          constexpr int j0 = (i * 8) / NM, j1 = ((i + 1) * 8) / NM;
          attn3_static_for<j0, j1>([&](auto j_c) __attribute__((always_inline)) {
            constexpr int j = decltype(j_c)::value, R = A3_S(j >> 1, qp, (j & 1) * 2);
            asm volatile("v_max3_f32 v[%c0], v[%c0], v[%c1], v[%c2]" ::"n"(A3_M(qp)), "n"(R), "n"(R + 1));
          });
        }
        if constexpr (LASTG && !(A3X & 2) && i >= NM - 4) {   // its own maxima: score tile kt was completed by MFMA NQK - 4 + kt, >= 4 MFMAs ago
          constexpr int kt = i - (NM - 4), R = A3_S(kt, q, 0);
          asm volatile("v_max3_f32 v[%c0], v[%c0], v[%c1], v[%c2]\n\tv_max3_f32 v[%c0], v[%c0], v[%c3], v[%c4]" ::"n"(A3_M(q)), "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3));
        }
      });
    });
    A3_STAMP(1);  // the five MFMA groups
  };

  // ring bookkeeping at the top of step t: tile t+1 landed for everyone (its K is read in this step), the slot of tile t-1 -- last read
  // in step t-1 -- refilled with tile t+NS-1
  auto sync_top = [&](int t) __attribute__((always_inline)) {
    if constexpr ((ABL & 64) == 0) {
      const int younger = ntiles - 3 - t;   // wave-uniform: tiles requested after t+2 (at most NS - 4 = 4 of them are in flight)
      if (younger >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * 4) : "memory");        // steady state
      else if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * 2) : "memory");   // (the last steps wait a little early: two rungs instead of four branches)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    static_assert(NS == 8, "the wait ladder is written for 4 younger tiles");
    A3_STAMP(2);  // landed-wait states + tail maxima + the tile DMA wait
    if constexpr (!(ABL & 128)) __builtin_amdgcn_s_barrier();
    A3_STAMP(3);  // barrier
    if constexpr ((ABL & 64) == 0) {
      if (t + NS - 1 < ntiles) issue_tile((t + NS - 1) & (NS - 1));
    }
    A3_STAMP(4);  // tile DMA issue
    if (t + 2 >= tfin) {   // (rare: the last one or two steps)
      for (int tile = (t == 0 ? 1 : t + 2); tile <= t + 2; ++tile)
        if (needs_finish(tile)) {
          finish_last_tile(tile, tile & (NS - 1));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
    }
  };
  // request the fragments step t consumes (loop entry / re-entry; inside the loop every step requests its successor's)
  auto prime = [&](int t) __attribute__((always_inline)) {
    const int tk = min(t + 1, ntiles - 1);
    if (t & 1) {
      read_vf((unsigned)(t & (NS - 1)), std::integral_constant<int, 1>{});
      read_kf((unsigned)(tk & (NS - 1)), std::integral_constant<int, 1>{});
    } else {
      read_vf((unsigned)(t & (NS - 1)), std::integral_constant<int, 0>{});
      read_kf((unsigned)(tk & (NS - 1)), std::integral_constant<int, 0>{});
    }
  };
  // Everything the fast loop reads is "touched" in front of it: a compiler-visible load still in flight at loop entry (a spill reload
  // of the cold code) turns into `s_waitcnt vmcnt(N)` ladders ending in vmcnt(0) INSIDE the loop -- executed every step, and the
  // hardware counter they wait on is the one the tile DMA lives on: the ring was drained once per step (seen in the ISA).
  auto touch_live_ins = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) asm volatile("" : "+v"(kbase[kc]));
#pragma unroll
    for (int c = 0; c < 2; ++c) asm volatile("" : "+v"(vbase[c]));
#pragma unroll
    for (int j = 0; j < PW; ++j) asm volatile("" : "+v"(dma_off[j]));
  };

  const std::true_type Tt{};
  const std::false_type Ff{};
  const std::integral_constant<int, 0> P0{};
  const std::integral_constant<int, 1> P1{};
  if (wave_active) {
    // S(0) against reference 0, its exact row maxima become the references
    qk_owned(0u);
    max_owned();
    sync_top(0);
    move_refs(Tt, Tt);
    int t = 0;
    const int nqk = ntiles - 1;   // steps that also produce their successor's scores: tiles 0 .. ntiles - 2; the last tile is consumed by the drain
    if (nqk > 0) {
      // ONE step body per parity, the rare reference move OUTSIDE the fast loop (see the header)
#ifdef A3_STAMPS
      for (int k = 0; k < 8; ++k) stamp_acc[k] = 0;
      stamp_last = __builtin_readcyclecounter();
      stamp_acc[5] = stamp_last - stamp_t0;     // prologue
#endif
      A3_WGSTAMP(1);
      A3_WGCYC(4);
      for (;;) {
        touch_live_ins();
        prime(t);
        soft0_now();
        bool done = false;
#pragma clang loop unroll(disable)
        for (;;) {
          if (t & 1) step(t, Tt, P1);
          else step(t, Tt, P0);
          ++t;
          if (t >= nqk) { done = true; break; }
          sync_top(t);
          if (must_move()) break;
        }
        if (done) break;
        move_refs(Ff, Ff);              // some score left the window of its reference: move it, recompute S(t) from the ring, re-enter
        qk_owned((unsigned)(t & (NS - 1)));
      }
      // t == ntiles - 1: its scores are there, query tile 0 already exponentiated (by the last step); nothing left to wait for or to request
      if (must_move()) {
        move_refs(Ff, Ff);
        qk_owned((unsigned)(t & (NS - 1)));
        prime(t);                       // (qk_owned used fragment set 0)
        soft0_now();
      }
    } else {
      prime(0);
      soft0_now();
    }
    if (rem > 0) mask_last_ones();
    // drain: softmax of query tiles 1 .. 4 + O^T += V^T P^T + row sums of the last tile
    if (t & 1) step(t, Ff, P1);
    else step(t, Ff, P0);
    A3_MFMA_LANDED();
  } else {
    const int nsync = ntiles > 1 ? ntiles - 1 : 1;
    for (int t = 0; t < nsync; ++t) sync_top(t);
  }

#ifdef A3_STAMPS
  const long long stamp_loop_end = __builtin_readcyclecounter();
#endif
  A3_WGSTAMP(2);
  A3_WGCYC(5);
  float lq[QT], mq[QT];
  attn3_static_for<0, QT>([&](auto q_c) __attribute__((always_inline)) {
    constexpr int q = decltype(q_c)::value;
    (void)&lq; (void)&mq;
    asm volatile("v_mov_b32 %0, v[%c2]\n\tv_mov_b32 %1, v[%c3]" : "=v"(lq[q]), "=v"(mq[q]) : "n"(A3_AL(q)), "n"(A3_M(q)));
  });
  if (p.stat_max && wave_active) {   // largest row maximum (natural units) of this wave's queries: m_ref + the relative running maximum
    float m = mref[0] + mq[0];
#pragma unroll
    for (int q = 1; q < QT; ++q) m = fmaxf(m, mref[q] + mq[q]);
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) m = fmaxf(m, __shfl_xor(m, sh, 64));   // (the maxima are per lane: all 64 lanes)
    const int mi = attn_ordered_int(m * 0.6931471805599453f);
    if (lane == 0 && mi > __atomic_load_n(p.stat_max, __ATOMIC_RELAXED)) atomicMax(p.stat_max, mi);
  }

  // ---- normalise, transpose through the (idle) ring, store whole rows ----
  constexpr int SP = DH + 8, PPR = DH / 8, NPC = QT * 16 * PPR / 64;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // every wave is done reading the last tile
  h16_t* stw = smem + wid * (QT * 16) * SP;
  f32x4 oq[QT][DVT];
  attn3_static_for<0, QT>([&](auto q_c) __attribute__((always_inline)) {
    attn3_static_for<0, DVT>([&](auto d_c) __attribute__((always_inline)) {
      constexpr int q = decltype(q_c)::value, dv = decltype(d_c)::value, R = A3_AO(q, dv, DVT);
      (void)&oq;
      float x0, x1, x2, x3;
      asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\tv_accvgpr_read_b32 %2, a[%c6]\n\tv_accvgpr_read_b32 %3, a[%c7]"
                   : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3) : "n"(R), "n"(R + 1), "n"(R + 2), "n"(R + 3));
      oq[q][dv] = f32x4{x0, x1, x2, x3};
    });
  });
#pragma unroll
  for (int q = 0; q < QT; ++q) {
    float l = lq[q];
    const float inv = 1.0f / l;
#pragma unroll
    for (int dv = 0; dv < DVT; ++dv) {
      const f32x4 v = oq[q][dv];
      *reinterpret_cast<h16x4*>(stw + (q * 16 + l15) * SP + dv * 16 + g * 4) =
          h16x4{(h16_t)(v[0] * inv), (h16_t)(v[1] * inv), (h16_t)(v[2] * inv), (h16_t)(v[3] * inv)};
    }
  }
  h16_t* Ob = reinterpret_cast<h16_t*>(p.O) + (int64_t)seq * p.o_seq_stride + head * DH;
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int pc = lane + 64 * i, row = pc / PPR, part = pc % PPR;
    const h16x8 v = *reinterpret_cast<const h16x8*>(stw + row * SP + part * 8);
    const int qi = q0 + row;
    if (qi < p.Tq) *reinterpret_cast<h16x8*>(Ob + (int64_t)qi * p.ldo + part * 8) = v;
  }
#ifdef A3_WGSTAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the output stores have left)
  A3_WGSTAMP(3);
  if (threadIdx.x == 0 && p.kv_slot) {
    long long* dbg = (long long*)p.kv_slot + (int64_t)blockIdx.x * 8;
    for (int k = 0; k < 6; ++k) dbg[k] = wg_t[k];
  }
#endif
#ifdef A3_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0 && p.kv_slot) {
    stamp_acc[7] = __builtin_readcyclecounter() - stamp_loop_end;   // epilogue
    long long* dbg = (long long*)p.kv_slot;
    for (int k = 0; k < 8; ++k) dbg[k] = stamp_acc[k];
  }
#endif
}
