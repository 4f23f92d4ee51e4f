// Flash-style multi-head attention for the FiLM denoiser (nn.MultiheadAttention as called at
// model/modules/transformer_modules.py:239-246,254-261; no masks, eval mode).
//
// Everything is computed TRANSPOSED so that softmax statistics and the P operand stay lane-local:
//   S^T[key][q]  = K Q^T      (A = K tile from LDS, B = Q fragments held in registers)
//   O^T[dv][q]   = V^T P^T    (A = V^T tile from LDS, B = P^T = the S^T accumulator registers)
// With the 16x16 MFMA C layout (col = lane&15 = query, row = 4*(lane>>4)+reg = key / dv) a lane owns
// one query column: running max / sum need only 2 cross-lane steps (xor 16, 32) and the
// exponentiated scores feed the second MFMA without any data movement.  V is therefore kept
// transposed in HBM ([.., head_dim, keys]) by the producing GEMM epilogue (EPI_STORE_T).
//
// Keys come from a "main" cache (per-sequence slot; the unconditional branch of classifier-free
// guidance shares one batch-invariant slot) plus an optional per-sample "tail" (the two
// time tokens that change every step, model/diffusion.py:386,392).
//
// Staging: bf16 tiles go HBM -> LDS with global_load_lds (16 B per lane) into a 2-deep ring, the
// next 64-key tile in flight while the current one is consumed; rows are unpadded and XOR-swizzled
// on the source address + on the read (conflict-free ds_read_b128 / ds_read_b64).  fp32 (parity
// mode) stages through registers into padded rows.
#pragma once
#include <type_traits>

#include "a2p_common.h"

struct AttnP {
  const void* Q;   // [nseq][Tq][ldq]  (+ head*DH)
  const void* K;   // [slot][keys][ldk]
  const void* VT;  // [slot][d][ldvt]
  void* O;         // [nseq][Tq][ldo]
  int64_t q_seq_stride, ldq;
  int64_t k_slot_stride, ldk;
  int64_t vt_slot_stride, ldvt;
  int64_t o_seq_stride, ldo;
  const float* ktail;  // [sample][s_tail][tail_row_stride] fp32 (+ head*DH), may be NULL
  const float* vtail;
  int64_t tail_sample_stride, tail_row_stride;
  const int* kv_slot;  // per-sequence slot index, NULL -> slot = seq
  int slot_rule, slot_b;  // != 0: the table's contents as a formula (no dependent load in front of the first K/V request):
                          // 1 = 1 + seq (conditional pass), 2 = 0 (unconditional), 3 = seq < slot_b ? 1 + seq : 0 (guidance: both)
  int tail_mod;        // sample = seq % tail_mod
  int kv_stream;       // 1: K/V slots other than 0 are read once per step (cached audio K/V): DMA them non-temporal so that
                       // they do not evict the chain kernels' weight streams from L2 / MALL (docs/lab_notebook_r1_r4.md section 4.2)
  int Tq, S_main, S_tail;
  float scale_log2e;   // log2(e) / sqrt(head_dim)
  // Workgroup -> (query block, head, sequence).  The grid is 1-D; with xcd_remap the 8 XCDs (block b runs on XCD b % 8) each
  // take whole (sequence, head) pairs, pair % 8 == xcd, and walk their query blocks back to back: all query blocks of a pair
  // -- and, with 8 heads, every sequence of a head, i.e. all users of the shared unconditional K/V slot -- read their K/V
  // through ONE L2 instead of five (T = 600: 5 query blocks per pair, each of which used to pull the pair's K/V from HBM)
  int nq, nheads, nseq, xcd_remap;
  // optional (a2p_attention_logit_max): the largest row maximum of the scaled scores q.k / sqrt(head_dim) any query of the launch
  // saw, as an order-preserving int (attn_ordered_int), atomicMax-ed once per wave at the end of the kernel.  The operand
  // rounding of the 16-bit modes turns into a logit error proportional to the logits' magnitude: this is the number that says
  // whether a checkpoint sits inside the range the parity tests cover (INTEGRATION.md "validity envelope").
  int* stat_max;
};

__device__ __forceinline__ int attn_ordered_int(float v) {   // monotone float -> int map (atomicMax on ints)
  const int i = __float_as_int(v);
  return i >= 0 ? i : i ^ 0x7fffffff;
}

__device__ __forceinline__ int attn_slot(const AttnP& p, int seq) {
  if (p.slot_rule == 1) return 1 + seq;
  if (p.slot_rule == 2) return 0;
  if (p.slot_rule == 3) return seq < p.slot_b ? 1 + seq : 0;
  return p.kv_slot ? p.kv_slot[seq] : seq;
}

// LDS tile geometry: K tile [64 keys][DH], V^T tile [DH][64 keys]
template <typename T, int DH>
struct AttnLds {
  static constexpr bool DMA = sizeof(T) == 2;
  static constexpr int KV = 64;
  static constexpr int NBUF = DMA ? 2 : 1;
  static constexpr int LSK = DMA ? DH : DH + 2;
  static constexpr int LSV = DMA ? KV : KV + 2;
  static constexpr int KSZ = KV * LSK, VSZ = DH * LSV;
  // swizzle of the 16-byte chunk index (bf16 only)
  // bf16: S^T tile kt takes the keys with (key % 8) / 4 == (kt & 1) of 32-key chunk kt / 2, so that a lane's P fragment of
  // a chunk (accumulator rows g*4..g*4+3 of tiles 2c and 2c+1) covers the 8 CONTIGUOUS keys 32c + 8g .. +7 and the
  // matching V^T fragment is one aligned ds_read_b128.  krow = K-tile row (key) feeding A-operand row i of tile kt.
  __device__ static __forceinline__ int krow(int kt, int i) {
    if constexpr (DMA) return 32 * (kt >> 1) + 8 * (i >> 2) + 4 * (kt & 1) + (i & 3);
    else return kt * 16 + i;
  }
  // 16-byte chunk swizzle of the K tile: the 16 rows {8j + 4b + r} read by one ds_read_b128 group must hit 16 distinct
  // 16-byte bank slots (slot = (row * row_bytes + pos * 16) mod 256)
  __device__ static constexpr __forceinline__ int kswz(int row) {
    if constexpr (DH == 64) return (((row >> 3) & 3) << 1) | ((row >> 1) & 1);
    else return (row >> 3) & 3;
  }
  __device__ static constexpr __forceinline__ int vswz(int row) { return (row >> 1) & 7; }
  __device__ static __forceinline__ int kidx(int row, int c) {
    if constexpr (DMA) return row * LSK + ((((c >> 3) ^ kswz(row)) << 3) | (c & 7));
    else return row * LSK + c;
  }
  __device__ static __forceinline__ int vidx(int row, int k) {
    if constexpr (DMA) return row * LSV + ((((k >> 3) ^ vswz(row)) << 3) | (k & 7));
    else return row * LSV + k;
  }
};

template <int AUX>  // cache policy of the tile DMA: 0 default, 2 non-temporal
__device__ __forceinline__ void attn_glds16(const h16_t* gsrc, h16_t* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, AUX);
}

// Cross-row (lane ^ 16, lane ^ 32) max-reduction of a per-lane value without the LDS crossbar: gfx950's
// v_permlane32_swap / v_permlane16_swap exchange whole 32- / 16-lane halves between two registers in the VALU, where
// __shfl_xor goes through ds_bpermute_b32 (an LDS round trip of ~100 cycles, four of them serialised per 64-key tile).
__device__ __forceinline__ float attn_rowgroup_max(float v) {
  // plain v_max_f32 through inline asm: fmaxf() on the bit-cast halves made hipcc canonicalise both operands first (4 extra VALU
  // instructions per call; the scores are finite or -inf here, never signalling NaNs)
  const unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // a[0] = {lo, lo}, a[1] = {hi, hi}
  unsigned w;
  asm("v_max_f32 %0, %1, %2" : "=v"(w) : "v"(a[0]), "v"(a[1]));
  const auto b = __builtin_amdgcn_permlane16_swap(w, w, false, false);   // rows {0,0,2,2} and {1,1,3,3}
  unsigned r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(b[0]), "v"(b[1]));
  return __uint_as_float(r);
}

// ABL: ablation switches for scratch/attn_bench.hip only (the library instantiates ABL = 0):
//   1 = no exp, 2 = no running-max reduction, 4 = no staging / barriers after tile 0, 8 = no PV MFMAs, 16 = no QK^T MFMAs,
//   32 = per-tile barrier without the DMA wait, 64 = per-tile DMA wait without the barrier (timing only: results are wrong)
// NWV: waves per workgroup (each wave owns QT*16 = 32 queries); 4 in production.  NWV = 2 was tried for its even grid (T = 600: 10
// query blocks per (sequence, head), B=8: 1280 workgroups = exactly 5 per CU instead of 2 or 3) and lost clearly -- cross attention
// 97 vs 70 us: every K/V tile then feeds 64 instead of 128 queries and the tile DMA / LDS traffic per query doubles.
// waves per SIMD the register allocation is held to: 3 for the 16-bit and the fp32 dh <= 64 kernels.  The fp32 dh = 128 kernel (the
// lip regressor of the audio front end: once per clip, 4 heads; 67 KB of LDS per workgroup) asks for 1: at 2 it spills 59 registers,
// and a request of 3 was silently dropped by hipcc (the code it compiled was the occupancy-1 code all along)
#ifndef A2P_ATTN_F32_MINWAVES
#define A2P_ATTN_F32_MINWAVES 3   // fp32 parity mode, head dim 64: 3 waves per SIMD spill 42 registers; 2 waves (208 registers, no spills) measured SLOWER: 96.6 vs 101.4 steps/s (profiles/r05_fp32_attn_occupancy_ab.txt)
#endif
template <typename T, int DH>
constexpr int attn_min_waves() { return (sizeof(T) == 4 && DH == 128) ? 1 : (sizeof(T) == 4 && DH == 64) ? A2P_ATTN_F32_MINWAVES : 3; }

template <typename T, int DH, int ABL = 0, int NWV = 4>
__global__ __launch_bounds__(64 * NWV, (attn_min_waves<T, DH>())) void attn_kernel(AttnP p) {
  using P = Prec<T>;
  using L = AttnLds<T, DH>;
  constexpr int QT = 2, BQ = NWV * QT * 16, KV = 64, NTHR = 64 * NWV;
  static_assert(NWV == 4 || sizeof(T) == 2, "the register-staged fp32 path is written for 4 waves");
  constexpr int KC = DH / P::KCH;   // k-chunks over head_dim (QK^T)
  constexpr int DVT = DH / 16;      // 16-row tiles of O^T
  constexpr int VEC = 16 / sizeof(T);
  __shared__ __attribute__((aligned(16))) T smem[L::NBUF * (L::KSZ + L::VSZ)];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  int qb, head, seq;
  {
    const int b = blockIdx.x;
    if (p.xcd_remap) {  // pairs = nheads * nseq is a multiple of 8 (checked by the host)
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
  const int q0 = qb * BQ + wid * (QT * 16);
  // waves of the last query block whose whole range lies past Tq (T=600: 1 of 20 waves) still take part in the tile DMA and
  // the barriers but skip the MFMA / softmax work: their issue slots go to the co-resident waves
  const bool wave_active = __builtin_amdgcn_readfirstlane(q0) < p.Tq;
  const int slot = attn_slot(p, seq);
  const bool kv_nt = p.kv_stream && slot != 0;  // slot 0 (null conditioning) is shared by the whole unconditional half: keep it cached
  const int S_total = p.S_main + p.S_tail;
  const int ntiles = (S_total + KV - 1) / KV;

  const T* Qb = reinterpret_cast<const T*>(p.Q) + (int64_t)seq * p.q_seq_stride + head * DH;
  const T* Kb = reinterpret_cast<const T*>(p.K) + (int64_t)slot * p.k_slot_stride + head * DH;
  const T* Vb = reinterpret_cast<const T*>(p.VT) + (int64_t)slot * p.vt_slot_stride + (int64_t)head * DH * p.ldvt;

  // Q fragments (B operand): lane (q = l15, k-group g)
  typename P::Frag qf[QT][KC];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    int q = q0 + qt * 16 + l15;
    if (q >= p.Tq) q = p.Tq - 1;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) qf[qt][kc] = P::load(Qb + (int64_t)q * p.ldq + kc * P::KCH + g * P::EPL);
  }

  f32x4 o[QT][DVT];
  float mrun[QT], lsum[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    mrun[qt] = -INFINITY;
    lsum[qt] = 0.f;
#pragma unroll
    for (int dv = 0; dv < DVT; ++dv) o[qt][dv] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- staging ------------------------------------------------------------------------------
  // bf16: source offsets of this wave's DMA pieces.  K piece pi = j*NWV + wid covers KRPI key rows, V^T piece pi covers 8 head-dim
  // rows.  The swizzle of a row is the XOR of a part that depends on the piece and a part that depends on the lane's row inside the
  // piece (kswz / vswz take disjoint bits from the two), so one per-lane offset + one per-lane swizzle serve all pieces: the piece
  // part is wave-uniform (scalar) arithmetic at issue time, not a register per piece.
  constexpr int CPR_ = DH / 8, KRPI_ = 64 / CPR_;
  constexpr int KPW = L::DMA ? CPR_ / NWV : 1, VPW = L::DMA ? DH / 8 / NWV : 1;
  const int widu = __builtin_amdgcn_readfirstlane(wid);
  [[maybe_unused]] int kbase = 0, ksw = 0, vbase = 0, vsw = 0;
  if constexpr (L::DMA) {
    const int kr0 = lane / CPR_, vr0 = lane >> 3;
    kbase = kr0 * (int)p.ldk;
    ksw = ((lane % CPR_) ^ L::kswz(kr0)) << 3;
    vbase = vr0 * (int)p.ldvt;
    vsw = ((lane & 7) ^ L::vswz(vr0)) << 3;
  }
  auto kpiece = [&](int j) __attribute__((always_inline)) {   // element offset of K piece j of this wave inside a tile
    const int pi = j * NWV + widu;
    return (int64_t)(pi * KRPI_) * p.ldk + (kbase + (ksw ^ (L::kswz(pi * KRPI_) << 3)));
  };
  auto vpiece = [&](int j) __attribute__((always_inline)) {
    const int pi = j * NWV + widu;
    return (int64_t)(pi * 8) * p.ldvt + (vbase + (vsw ^ (L::vswz(pi * 8) << 3)));
  };
  auto stage_dma = [&](int tile, int buf) {
    if constexpr (L::DMA) {
      h16_t* Ks = reinterpret_cast<h16_t*>(smem) + buf * (L::KSZ + L::VSZ);
      h16_t* Vs = Ks + L::KSZ;
      const h16_t* Kt = reinterpret_cast<const h16_t*>(Kb) + (int64_t)tile * KV * p.ldk;
      const h16_t* Vt = reinterpret_cast<const h16_t*>(Vb) + tile * KV;
      if (kv_nt) {  // block-uniform
#pragma unroll
        for (int j = 0; j < KPW; ++j) attn_glds16<2>(Kt + kpiece(j), Ks + (j * NWV + widu) * KRPI_ * DH);
#pragma unroll
        for (int j = 0; j < VPW; ++j) attn_glds16<2>(Vt + vpiece(j), Vs + (j * NWV + widu) * 8 * KV);
      } else {
#pragma unroll
        for (int j = 0; j < KPW; ++j) attn_glds16<0>(Kt + kpiece(j), Ks + (j * NWV + widu) * KRPI_ * DH);
#pragma unroll
        for (int j = 0; j < VPW; ++j) attn_glds16<0>(Vt + vpiece(j), Vs + (j * NWV + widu) * 8 * KV);
      }
    }
  };
  auto stage_regs = [&](int tile) {  // fp32: global -> registers -> padded LDS
    T* Ks = smem;
    T* Vs = smem + L::KSZ;
    const int kv0 = tile * KV;
    constexpr int KVR = DH / VEC, VVR = KV / VEC;
    for (int v = tid; v < KV * KVR; v += 256) {
      const int r = v / KVR, c = v % KVR;
      const uint4 val = *reinterpret_cast<const uint4*>(Kb + (int64_t)(kv0 + r) * p.ldk + c * VEC);
      reinterpret_cast<uint2*>(&Ks[r * L::LSK + c * VEC])[0] = make_uint2(val.x, val.y);
      reinterpret_cast<uint2*>(&Ks[r * L::LSK + c * VEC])[1] = make_uint2(val.z, val.w);
    }
    for (int v = tid; v < DH * VVR; v += 256) {
      const int r = v / VVR, c = v % VVR;
      uint4 val = *reinterpret_cast<const uint4*>(Vb + (int64_t)r * p.ldvt + kv0 + c * VEC);
      // columns of keys past the end were never written (0 x nan = nan in the PV product): zero them (see load_vfr)
      const int nvalid = p.S_main - (kv0 + c * VEC);   // (the time-token tail rows are patched in afterwards)
      if (nvalid < 4) {
        if (nvalid < 1) val.x = 0u;
        if (nvalid < 2) val.y = 0u;
        if (nvalid < 3) val.z = 0u;
        val.w = 0u;
      }
      reinterpret_cast<uint2*>(&Vs[r * L::LSV + c * VEC])[0] = make_uint2(val.x, val.y);
      reinterpret_cast<uint2*>(&Vs[r * L::LSV + c * VEC])[1] = make_uint2(val.z, val.w);
    }
  };

  // One 64-key tile.  BUF (LDS ring slot) and MASKED (tile reaches past the last key) are compile-time so the
  // fragment addresses fold into immediates and the -inf masking costs nothing on full tiles (this loop is
  // instruction-issue bound: ~250 VALU + 32 MFMA per tile per wave before, 3 waves per SIMD).
  auto tile_body = [&](int tile, auto buf_c, auto masked_c, auto active_c) __attribute__((always_inline)) {
    constexpr int BUF = decltype(buf_c)::value;
    constexpr bool MASKED = decltype(masked_c)::value;
    constexpr bool ACTIVE = decltype(active_c)::value;
    const int kv0 = tile * KV;
    constexpr int buf = L::DMA ? BUF : 0;
    T* Ks = smem + buf * (L::KSZ + L::VSZ);
    T* Vs = Ks + L::KSZ;
    if (!(ABL & 4) || tile == 0) {
      // bf16: tile landed (vmcnt(0)) and the other buffer is free; fp32: previous tile consumed
      if constexpr ((ABL & 32) != 0) {          // ablation: barrier, but no wait for the tile DMA
        if (tile == 0) __syncthreads();
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      } else if constexpr ((ABL & 64) != 0) {   // ablation: wait for this wave's DMA pieces, but no barrier
        if (tile == 0) __syncthreads();
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else {
        __syncthreads();
      }
      if constexpr (L::DMA) {
        if (tile + 1 < ntiles) stage_dma(tile + 1, buf ^ 1);
      } else {
        stage_regs(tile);
      }
    }
    if (p.S_tail > 0 && kv0 + KV > p.S_main) {  // block-uniform: patch the time-token rows into the tile
      if constexpr (!L::DMA) __syncthreads();
      const int sample = seq % p.tail_mod;
      for (int e = tid; e < p.S_tail * DH; e += NTHR) {
        const int j = e / DH, c = e % DH;
        const int kl = p.S_main + j - kv0;
        if (kl >= 0 && kl < KV) {
          const int64_t off = (int64_t)sample * p.tail_sample_stride + (int64_t)j * p.tail_row_stride + head * DH + c;
          Ks[L::kidx(kl, c)] = from_f32<T>(p.ktail[off]);
          Vs[L::vidx(c, kl)] = from_f32<T>(p.vtail[off]);
        }
      }
      __syncthreads();
    } else if constexpr (!L::DMA) {
      __syncthreads();
    }

    if constexpr (!ACTIVE) return;
    // ---- S^T = K Q^T for 4 key tiles x QT query tiles ----
    f32x4 s[4][QT];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) s[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const typename P::Frag kf = P::load(&Ks[L::kidx(L::krow(kt, l15), kc * P::KCH + g * P::EPL)]);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          if constexpr (!(ABL & 16)) s[kt][qt] = P::mfma(kf, qf[qt][kc], s[kt][qt]);
          else asm volatile("" : "+v"(s[kt][qt]) : "v"(kf));
        }
      }
    }
    // 16-bit path: all V^T fragments of the tile are requested NOW, behind the QK^T MFMAs, so that their LDS latency passes
    // under the softmax arithmetic (hipcc otherwise sinks each ds_read_b128 directly in front of its MFMA with an
    // s_waitcnt lgkmcnt(0) between them: 8 exposed LDS round trips per tile and wave)
    [[maybe_unused]] h16x8 vfr[2][DVT];
    auto load_vfr = [&](int c) __attribute__((always_inline)) {
#pragma unroll
      for (int dv = 0; dv < DVT; ++dv)
        vfr[c][dv] = *reinterpret_cast<const h16x8*>(&Vs[L::vidx(dv * 16 + l15, c * 32 + g * 8)]);
#ifndef A2P_ATTN_AB_R3   // (scratch A/B build: the round-3 tile body, to price the two round-4 additions on one box)
      if constexpr (MASKED) {
        // Keys past the end carry P = 0, but their V^T columns were never written by anybody: whatever the workspace held there
        // goes into the MFMA, and 0 x (inf | nan) = nan.  One earlier non-finite forward on the context (an overflowing checkpoint:
        // a2p_check_finite) would poison every later one through these columns: zero them.  Last tile only.
        const int nvalid = S_total - (kv0 + c * 32 + g * 8);   // keys of this lane's 8-key fragment that exist
#pragma unroll
        for (int dv = 0; dv < DVT; ++dv)
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (e >= nvalid) vfr[c][dv][e] = (h16_t)0.f;
      }
#endif
    };
    if constexpr (sizeof(T) == 2) load_vfr(0);   // the second half follows the softmax (register budget: 3 waves per SIMD)
    // ---- online softmax (log2 domain); lane owns query l15, keys kt*16 + g*4 + r ----
    // VALU diet (this loop is VALU-bound, not MFMA-bound): masking only on the last tile, the
    // 1/sqrt(dh)*log2(e) scale folded into the exp2 argument (one fma per score).
    if constexpr (MASKED) {
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
    for (int qt = 0; qt < QT; ++qt) {
      float mx = s[0][qt][0];
      if constexpr (!(ABL & 2)) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][qt][r]);
        mx = attn_rowgroup_max(mx);
      }
      const float mnew = fmaxf(mrun[qt], mx * p.scale_log2e);
      const float alpha = __builtin_amdgcn_exp2f(mrun[qt] - mnew);
      mrun[qt] = mnew;
      // two scores per VALU instruction where the ISA has a packed form (v_pk_fma_f32 / v_pk_add_f32): this loop is
      // instruction-issue bound (profiles/r01_attn_ablation.txt), the exp2 itself has no packed form
      f32x2 ps2 = {0.f, 0.f};
      const f32x2 sc2 = {p.scale_log2e, p.scale_log2e}, mn2 = {-mnew, -mnew};
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x2 v = {s[kt][qt][2 * h], s[kt][qt][2 * h + 1]};
          v = __builtin_elementwise_fma(v, sc2, mn2);
          if constexpr (!(ABL & 1)) {
            v[0] = __builtin_amdgcn_exp2f(v[0]);
            v[1] = __builtin_amdgcn_exp2f(v[1]);
          }
          ps2 += v;
          s[kt][qt][2 * h] = v[0];
          s[kt][qt][2 * h + 1] = v[1];
        }
      const float ps = ps2[0] + ps2[1];
      lsum[qt] = lsum[qt] * alpha + ps;
      // O rescale, unconditionally: alpha is exactly 1.0f when the running max did not move.  (Skipping it under a wave-uniform
      // branch cost more than it saved: hipcc copied all 32 accumulator registers on both paths, ~40 v_mov_b64 per tile.)
#pragma unroll
      for (int dv = 0; dv < DVT; ++dv)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[qt][dv][r] *= alpha;
    }
    // ---- O^T += V^T P^T ----
    if constexpr (sizeof(T) == 2) {
      load_vfr(1);
  #pragma unroll
      for (int c = 0; c < 2; ++c) {  // 32-key chunk: k-slot e of lane group g -> key c*32 + g*8 + e (see AttnLds::krow)
        h16x8 pf[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pf[qt][r] = (h16_t)s[2 * c][qt][r];
            pf[qt][4 + r] = (h16_t)s[2 * c + 1][qt][r];
          }
        }
#pragma unroll
        for (int dv = 0; dv < DVT; ++dv) {
          const h16x8 vf = vfr[c][dv];
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            if constexpr (!(ABL & 8)) o[qt][dv] = P::mfma(vf, pf[qt], o[qt][dv]);
            else asm volatile("" : "+v"(o[qt][dv]) : "v"(vf), "v"(pf[qt]));
          }
        }
      }
      } else {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // 4-key chunk: k-group g -> key kt*16 + g*4 + r
#pragma unroll
          for (int dv = 0; dv < DVT; ++dv) {
            const float vf = Vs[L::vidx(dv * 16 + l15, kt * 16 + g * 4 + r)];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) o[qt][dv] = P::mfma(vf, s[kt][qt][r], o[qt][dv]);
          }
        }
    }
  };
  using std::integral_constant;
  if constexpr (L::DMA) stage_dma(0, 0);
  // the wave_active test is hoisted out of the tile loop (two loop nests): inside it, the skipped-compute path made every
  // accumulator a phi of "updated" and "untouched" and hipcc paid ~20 v_mov_b64 per tile for it
  auto tile_loop = [&](auto active_c) __attribute__((always_inline)) {
    for (int tile = 0; tile < ntiles; tile += 2) {
      const bool last0 = tile + 1 >= ntiles;
      if (last0 && ntiles * KV > S_total) tile_body(tile, integral_constant<int, 0>{}, integral_constant<bool, true>{}, active_c);
      else tile_body(tile, integral_constant<int, 0>{}, integral_constant<bool, false>{}, active_c);
      if (last0) break;
      if (tile + 2 >= ntiles && ntiles * KV > S_total) tile_body(tile + 1, integral_constant<int, 1>{}, integral_constant<bool, true>{}, active_c);
      else tile_body(tile + 1, integral_constant<int, 1>{}, integral_constant<bool, false>{}, active_c);
    }
  };
  if (wave_active) tile_loop(integral_constant<bool, true>{});
  else tile_loop(integral_constant<bool, false>{});

#ifndef A2P_ATTN_AB_R3
  if (p.stat_max && wave_active) {   // largest row maximum (natural units) of this wave's queries
    float m = mrun[0];
#pragma unroll
    for (int qt = 1; qt < QT; ++qt) m = fmaxf(m, mrun[qt]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));   // over the 16 queries of a lane group (rows are already reduced)
    // One address for the whole launch: an unconditional atomic per wave serialises ~5000 of them at one L2 channel (measured: the
    // body model's attention launches went from 31 / 47 us to 65 / 77 us).  The stored maximum only ever grows, so a plain load filters
    // all but the few waves that actually raise it (a stale read costs one redundant atomic, never a wrong result).
    const int mi = attn_ordered_int(m * 0.6931471805599453f);   // log2 domain -> natural units
    if (lane == 0 && mi > __atomic_load_n(p.stat_max, __ATOMIC_RELAXED)) atomicMax(p.stat_max, mi);
  }
#endif
  // ---- normalise and store: lane owns query l15, rows dv*16 + g*4 + {0..3} ----
  if constexpr (sizeof(T) == 2) {
    // 16-bit: a lane holds 4 consecutive head-dim values (8 bytes) of one query, so a direct store writes 16 rows x 32 bytes
    // per instruction -- rows of less than 64 contiguous bytes are the slow store regime (scratch/issue_probe: ~9 cycles per
    // touched line against ~3.5).  The wave's [32 queries][DH] block is transposed through its slice of the (now idle) K/V ring
    // instead and leaves as 16-byte pieces, DH/8 adjacent lanes per query: whole 128-byte rows per store at DH = 64.
    constexpr int SP = DH + 8;                      // padded row (elements): 2-way instead of 16-way bank conflicts on the writes
    constexpr int PPR = DH / 8;                     // 16-byte pieces per query row
    constexpr int NPC = QT * 16 * PPR / 64;         // pieces per lane
    static_assert(NWV * QT * 16 * SP <= L::NBUF * (L::KSZ + L::VSZ), "staging does not fit the K/V ring");
    __syncthreads();                                // every wave is done reading the last tile
    T* stw = smem + wid * (QT * 16 * SP);
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
    T* Ob = reinterpret_cast<T*>(p.O) + (int64_t)seq * p.o_seq_stride + head * DH;
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int pc = lane + 64 * i, row = pc / PPR, part = pc % PPR;
      const h16x8 v = *reinterpret_cast<const h16x8*>(stw + row * SP + part * 8);
      const int q = q0 + row;
      if (q < p.Tq) *reinterpret_cast<h16x8*>(Ob + (int64_t)q * p.ldo + part * 8) = v;
    }
  } else {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float l = lsum[qt];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      const float inv = 1.0f / l;
      const int q = q0 + qt * 16 + l15;
      if (q >= p.Tq) continue;
      T* Op = reinterpret_cast<T*>(p.O) + (int64_t)seq * p.o_seq_stride + (int64_t)q * p.ldo + head * DH;
#pragma unroll
      for (int dv = 0; dv < DVT; ++dv) {
        const f32x4 v = o[qt][dv];
        *reinterpret_cast<float4*>(Op + dv * 16 + g * 4) = make_float4(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Key-split attention for small forwards (16-bit modes; kernels_small.h's regime: BASELINE configs[0] is 2 sequences x 240
// frames).  attn_kernel gives such a forward 32 workgroups that each walk their 4-10 key tiles one after the other -- a serial
// chain of (tile load, QK^T, softmax, PV) round trips on an eighth of the chip, 12 us per launch.  Here a workgroup owns 32
// queries of one (sequence, head) and its four waves split the KEYS: wave w takes tiles w, w + 4, ... (one tile each at 240
// keys), with the fragments read straight from global memory into registers (no LDS ring, no barriers: nothing is shared between
// the waves but the queries), and the four partial (max, sum, O) states are merged through LDS at the end -- the usual
// log-sum-exp combine.  4x the workgroups, a quarter of the dependent tile steps.
// Same per-tile arithmetic as attn_kernel (S^T = K Q^T, lane-local online softmax, O^T += V^T P^T, fragment / key mapping of
// AttnLds::krow); the summation ORDER over keys differs (per-wave partial sums), so results agree with attn_kernel to fp32
// rounding of the softmax sums, not bit for bit.  Host contract: ldvt % 8 == 0 (16-byte V^T fragment loads), S_tail <= 2; any S_main.
// ------------------------------------------------------------------------------------------------
template <int DH, int NW = 4, int QT = 2>
__global__ __launch_bounds__(64 * NW) void attn_ksplit_kernel(AttnP p) {
  using T = h16_t;
  using P = Prec<T>;
  using L = AttnLds<T, DH>;
  constexpr int KV = 64;
  constexpr int KC = DH / P::KCH, DVT = DH / 16;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) float so[NW][QT][DVT][64][4];
  __shared__ float sm[NW][QT][16], sl[NW][QT][16];

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int b = blockIdx.x;
  const int qb = b % p.nq, head = (b / p.nq) % p.nheads, seq = b / (p.nq * p.nheads);
  const int q0 = qb * (QT * 16);
  const int slot = attn_slot(p, seq);
  const int S_total = p.S_main + p.S_tail;
  const int ntiles = (S_total + KV - 1) / KV;

  const T* Qb = reinterpret_cast<const T*>(p.Q) + (int64_t)seq * p.q_seq_stride + head * DH;
  const T* Kb = reinterpret_cast<const T*>(p.K) + (int64_t)slot * p.k_slot_stride + head * DH;
  const T* Vb = reinterpret_cast<const T*>(p.VT) + (int64_t)slot * p.vt_slot_stride + (int64_t)head * DH * p.ldvt;
  // time-token rows (at most two, host contract); with none, the K base serves as a harmless in-bounds address for the loads every
  // lane issues on the straight-line path below
  const int n_tail = p.S_tail;
  const int64_t tail_off = (int64_t)(seq % (p.tail_mod > 0 ? p.tail_mod : 1)) * p.tail_sample_stride + head * DH;
  const float* ksrc = n_tail > 0 ? p.ktail + tail_off : reinterpret_cast<const float*>(Kb);
  const float* vsrc = n_tail > 0 ? p.vtail + tail_off : reinterpret_cast<const float*>(Kb);
  const int64_t trs = n_tail > 0 ? p.tail_row_stride : 0;

  h16x8 qf[QT][KC];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    int q = q0 + qt * 16 + l15;
    if (q >= p.Tq) q = p.Tq - 1;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) qf[qt][kc] = P::load(Qb + (int64_t)q * p.ldq + kc * P::KCH + g * P::EPL);
  }
  f32x4 o[QT][DVT];
  float mrun[QT], lsum[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    mrun[qt] = -INFINITY;
    lsum[qt] = 0.f;
#pragma unroll
    for (int dv = 0; dv < DVT; ++dv) o[qt][dv] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // fragments of a tile that lies wholly inside the main cache: plain 16-byte loads
  auto load_main = [&](int kv0, h16x8(&kf)[4][KC], h16x8(&vf)[2][DVT]) __attribute__((always_inline)) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
        kf[kt][kc] = *reinterpret_cast<const h16x8*>(Kb + (int64_t)(kv0 + L::krow(kt, l15)) * p.ldk + kc * P::KCH + g * P::EPL);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int dv = 0; dv < DVT; ++dv)
        vf[c][dv] = *reinterpret_cast<const h16x8*>(Vb + (int64_t)(dv * 16 + l15) * p.ldvt + kv0 + c * 32 + g * 8);
  };
  // the tile(s) with the end of the main cache, the time tokens and the padding -- STRAIGHT-LINE code: this wave is the one the
  // merge waits for, and per-lane branches around element loads (or a uniform branch per fragment: hipcc drains vmcnt at each)
  // turn its one memory round trip into a dozen.  Every lane issues every load: main-cache fragments from a row / column clamped
  // into the cache (masked keys: finite real data for K; V^T's padding columns are unwritten memory and are bit-masked to zero),
  // the time-token rows from a clamped token index, and selects put them in place.
  auto load_edge = [&](int kv0, h16x8(&kf)[4][KC], h16x8(&vf)[2][DVT]) __attribute__((always_inline)) {
    // a lane's four key rows (one per S^T tile) lie 4 or more keys apart and the time tokens are adjacent keys: at most ONE of the
    // four is a time-token row, so one candidate row is loaded per lane (not one per tile: 48 registers less)
    int jsel = 0, ktsel = -1;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int kr = kv0 + L::krow(kt, l15);
      const int krc = kr < p.S_main ? kr : p.S_main - 1;
      const int j = kr - p.S_main;
      const bool is_tail = j >= 0 && j < n_tail;
      jsel = is_tail ? j : jsel;
      ktsel = is_tail ? kt : ktsel;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
        kf[kt][kc] = *reinterpret_cast<const h16x8*>(Kb + (int64_t)krc * p.ldk + kc * P::KCH + g * P::EPL);
    }
    float4 kt0[KC], kt1[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const float4* tp = reinterpret_cast<const float4*>(ksrc + jsel * trs + kc * P::KCH + g * P::EPL);
      kt0[kc] = tp[0];
      kt1[kc] = tp[1];
    }
    u32x4 vm[2][DVT];
    float tv[2][DVT];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int kk = kv0 + c * 32 + g * 8;
      const int kkc = kk <= (int)p.ldvt - 8 ? kk : (int)p.ldvt - 8;
#pragma unroll
      for (int dv = 0; dv < DVT; ++dv) vm[c][dv] = *reinterpret_cast<const u32x4*>(Vb + (int64_t)(dv * 16 + l15) * p.ldvt + kkc);
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int dv = 0; dv < DVT; ++dv) tv[jj][dv] = vsrc[(jj < n_tail ? jj : 0) * trs + dv * 16 + l15];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const float4 t0 = kt0[kc], t1 = kt1[kc];
      const h16x8 tf = {from_f32<T>(t0.x), from_f32<T>(t0.y), from_f32<T>(t0.z), from_f32<T>(t0.w),
                        from_f32<T>(t1.x), from_f32<T>(t1.y), from_f32<T>(t1.z), from_f32<T>(t1.w)};
      const u32x4 a = __builtin_bit_cast(u32x4, tf);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const u32x4 m = __builtin_bit_cast(u32x4, kf[kt][kc]);
        u32x4 r;
#pragma unroll
        for (int d2 = 0; d2 < 4; ++d2) r[d2] = kt == ktsel ? a[d2] : m[d2];
        kf[kt][kc] = __builtin_bit_cast(h16x8, r);
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int kk = kv0 + c * 32 + g * 8;
      const int nvalid = p.S_main - kk;                  // elements [0, nvalid) of the group are main-cache keys
      const int e0 = n_tail > 0 ? p.S_main - kk : -1, e1 = n_tail > 1 ? p.S_main + 1 - kk : -1;   // where the time tokens go
#pragma unroll
      for (int dv = 0; dv < DVT; ++dv) {
        const unsigned b0 = (unsigned)__builtin_bit_cast(unsigned short, from_f32<T>(tv[0][dv]));
        const unsigned b1 = (unsigned)__builtin_bit_cast(unsigned short, from_f32<T>(tv[1][dv]));
        u32x4 f = vm[c][dv];
#pragma unroll
        for (int d2 = 0; d2 < 4; ++d2) {
          const unsigned msk = nvalid >= 2 * d2 + 2 ? 0xffffffffu : (nvalid == 2 * d2 + 1 ? 0x0000ffffu : 0u);
          unsigned w = f[d2] & msk;
          w = e0 == 2 * d2 ? ((w & 0xffff0000u) | b0) : w;
          w = e0 == 2 * d2 + 1 ? ((w & 0x0000ffffu) | (b0 << 16)) : w;
          w = e1 == 2 * d2 ? ((w & 0xffff0000u) | b1) : w;
          w = e1 == 2 * d2 + 1 ? ((w & 0x0000ffffu) | (b1 << 16)) : w;
          f[d2] = w;
        }
        vf[c][dv] = __builtin_bit_cast(h16x8, f);
      }
    }
  };
  // one 64-key tile: S^T = K Q^T, lane-local online softmax, O^T += V^T P^T (attn_kernel's arithmetic)
  auto consume = [&](int kv0, const h16x8(&kf)[4][KC], const h16x8(&vf)[2][DVT]) __attribute__((always_inline)) {
    f32x4 s[4][QT];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) s[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s[kt][qt] = P::mfma(kf[kt][kc], qf[qt][kc], s[kt][qt]);
    if (kv0 + KV > S_total) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool dead = kv0 + L::krow(kt, g * 4 + r) >= S_total;
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) s[kt][qt][r] = dead ? -INFINITY : s[kt][qt][r];
        }
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float mx = s[0][qt][0];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][qt][r]);
      mx = attn_rowgroup_max(mx);
      const float mnew = fmaxf(mrun[qt], mx * p.scale_log2e);
      const float alpha = __builtin_amdgcn_exp2f(mrun[qt] - mnew);
      mrun[qt] = mnew;
      float ps = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = __builtin_amdgcn_exp2f(fmaf(s[kt][qt][r], p.scale_log2e, -mnew));
          ps += v;
          s[kt][qt][r] = v;
        }
      lsum[qt] = lsum[qt] * alpha + ps;
#pragma unroll
      for (int dv = 0; dv < DVT; ++dv)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[qt][dv][r] *= alpha;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      h16x8 pf[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pf[qt][r] = (h16_t)s[2 * c][qt][r];
          pf[qt][4 + r] = (h16_t)s[2 * c + 1][qt][r];
        }
#pragma unroll
      for (int dv = 0; dv < DVT; ++dv)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[qt][dv] = P::mfma(vf[c][dv], pf[qt], o[qt][dv]);
    }
  };

  // wave w: tiles w, w + NW, ...; the NEXT tile's fragments are requested before the current tile is consumed (the loop is a chain
  // of memory round trips otherwise: ~2 us per tile against ~0.7 us of arithmetic).  Each arm of the (wave-uniform) branch holds its
  // loads AND the consume, so that the vmcnt in front of the MFMAs is exact for that arm instead of the join's conservative 0.
  if (wid < ntiles) {
    h16x8 kf[4][KC], vf[2][DVT];
    if (wid * KV + KV <= p.S_main) load_main(wid * KV, kf, vf);
    else load_edge(wid * KV, kf, vf);
    for (int tile = wid; tile < ntiles; tile += NW) {
      const int kv0 = tile * KV, nxt = tile + NW;
      if (nxt >= ntiles) {   // last tile of this wave: nothing to request
        consume(kv0, kf, vf);
        break;
      }
      h16x8 kn[4][KC], vn[2][DVT];
      if (nxt * KV + KV <= p.S_main) {
        load_main(nxt * KV, kn, vn);
        consume(kv0, kf, vf);
      } else {
        load_edge(nxt * KV, kn, vn);
        consume(kv0, kf, vf);
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) kf[kt][kc] = kn[kt][kc];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int dv = 0; dv < DVT; ++dv) vf[c][dv] = vn[c][dv];
    }
  }

  // ---- merge the NW key ranges: wave w normalises and stores the (query tile, head-dim tile) pairs w, w + NW, ... ----
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    float l = lsum[qt];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (g == 0) {
      sm[wid][qt][l15] = mrun[qt];
      sl[wid][qt][l15] = l;
    }
#pragma unroll
    for (int dv = 0; dv < DVT; ++dv) *reinterpret_cast<f32x4*>(&so[wid][qt][dv][lane][0]) = o[qt][dv];
  }
  __syncthreads();
  if (p.stat_max && wid == 0) {   // a2p_attention_logit_max: largest merged row maximum of this workgroup's queries (as attn_kernel)
    float m = -INFINITY;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int w = 0; w < NW; ++w) m = fmaxf(m, sm[w][qt][l15]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    const int mi = attn_ordered_int(m * 0.6931471805599453f);   // log2 domain -> natural units
    if (lane == 0 && mi > __atomic_load_n(p.stat_max, __ATOMIC_RELAXED)) atomicMax(p.stat_max, mi);
  }
  T* Ob = reinterpret_cast<T*>(p.O) + (int64_t)seq * p.o_seq_stride + head * DH;
  for (int pr = wid; pr < QT * DVT; pr += NW) {
    const int qt = pr / DVT, dv = pr % DVT;
    float m[NW], mall = -INFINITY;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      m[w] = sm[w][qt][l15];
      mall = fmaxf(mall, m[w]);
    }
    float lall = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float a = __builtin_amdgcn_exp2f(m[w] - mall);   // a wave without tiles: m = -inf -> weight 0
      lall += a * sl[w][qt][l15];
      const f32x4 ow = *reinterpret_cast<const f32x4*>(&so[w][qt][dv][lane][0]);
      acc += ow * a;
    }
    const float inv = 1.0f / lall;
    const int q = q0 + qt * 16 + l15;
    if (q < p.Tq)
      *reinterpret_cast<h16x4*>(Ob + (int64_t)q * p.ldo + dv * 16 + g * 4) =
          h16x4{(h16_t)(acc[0] * inv), (h16_t)(acc[1] * inv), (h16_t)(acc[2] * inv), (h16_t)(acc[3] * inv)};
  }
}
