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
  int tail_mod;        // sample = seq % tail_mod
  int Tq, S_main, S_tail;
  float scale_log2e;   // log2(e) / sqrt(head_dim)
};

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
  __device__ static __forceinline__ int kswz(int row) {
    if constexpr (DH == 64) return (row >> 1) & 7;
    else return (-(row >> 2)) & 3;
  }
  __device__ static __forceinline__ int vswz(int row) { return (row >> 1) & 7; }
  __device__ static __forceinline__ int kidx(int row, int c) {
    if constexpr (DMA) return row * LSK + ((((c >> 3) ^ kswz(row)) << 3) | (c & 7));
    else return row * LSK + c;
  }
  __device__ static __forceinline__ int vidx(int row, int k) {
    if constexpr (DMA) return row * LSV + ((((k >> 3) ^ vswz(row)) << 3) | (k & 7));
    else return row * LSV + k;
  }
};

__device__ __forceinline__ void attn_glds16(const bf16_t* gsrc, bf16_t* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <typename T, int DH>
__global__ __launch_bounds__(256, 3) void attn_kernel(AttnP p) {
  using P = Prec<T>;
  using L = AttnLds<T, DH>;
  constexpr int QT = 2, BQ = 4 * QT * 16, KV = 64;
  constexpr int KC = DH / P::KCH;   // k-chunks over head_dim (QK^T)
  constexpr int DVT = DH / 16;      // 16-row tiles of O^T
  constexpr int VEC = 16 / sizeof(T);
  __shared__ __attribute__((aligned(16))) T smem[L::NBUF * (L::KSZ + L::VSZ)];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int seq = blockIdx.z, head = blockIdx.y;
  const int q0 = blockIdx.x * BQ + wid * (QT * 16);
  const int slot = p.kv_slot ? p.kv_slot[seq] : seq;
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
  [[maybe_unused]] int64_t ksrc[2], vsrc[2];  // bf16: per-lane source offsets of this wave's DMA pieces
  if constexpr (L::DMA) {
    constexpr int CPR = DH / 8;          // 16-byte chunks per K row
    constexpr int KRPI = 64 / CPR;       // K rows per wave-instruction
    constexpr int KPW = CPR / 4;         // K instructions per wave (4 waves)
    constexpr int VPW = DH / 32;         // V^T instructions per wave (8 rows each)
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
      const int row = (j * 4 + wid) * KRPI + lane / CPR, pos = lane % CPR;
      ksrc[j] = (int64_t)row * p.ldk + ((pos ^ L::kswz(row)) << 3);
    }
#pragma unroll
    for (int j = 0; j < VPW; ++j) {
      const int row = (j * 4 + wid) * 8 + (lane >> 3), pos = lane & 7;
      vsrc[j] = (int64_t)row * p.ldvt + ((pos ^ L::vswz(row)) << 3);
    }
  }
  auto stage_dma = [&](int tile, int buf) {
    if constexpr (L::DMA) {
      constexpr int CPR = DH / 8, KRPI = 64 / CPR, KPW = CPR / 4, VPW = DH / 32;
      bf16_t* Ks = reinterpret_cast<bf16_t*>(smem) + buf * (L::KSZ + L::VSZ);
      bf16_t* Vs = Ks + L::KSZ;
      const bf16_t* Kt = reinterpret_cast<const bf16_t*>(Kb) + (int64_t)tile * KV * p.ldk;
      const bf16_t* Vt = reinterpret_cast<const bf16_t*>(Vb) + tile * KV;
#pragma unroll
      for (int j = 0; j < KPW; ++j) attn_glds16(Kt + ksrc[j], Ks + (j * 4 + wid) * KRPI * DH);
#pragma unroll
      for (int j = 0; j < VPW; ++j) attn_glds16(Vt + vsrc[j], Vs + (j * 4 + wid) * 8 * KV);
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
      const uint4 val = *reinterpret_cast<const uint4*>(Vb + (int64_t)r * p.ldvt + kv0 + c * VEC);
      reinterpret_cast<uint2*>(&Vs[r * L::LSV + c * VEC])[0] = make_uint2(val.x, val.y);
      reinterpret_cast<uint2*>(&Vs[r * L::LSV + c * VEC])[1] = make_uint2(val.z, val.w);
    }
  };

  if constexpr (L::DMA) stage_dma(0, 0);
  for (int tile = 0; tile < ntiles; ++tile) {
    const int kv0 = tile * KV;
    const int buf = L::DMA ? (tile & 1) : 0;
    T* Ks = smem + buf * (L::KSZ + L::VSZ);
    T* Vs = Ks + L::KSZ;
    __syncthreads();  // bf16: tile landed (vmcnt(0)) and the other buffer is free; fp32: previous tile consumed
    if constexpr (L::DMA) {
      if (tile + 1 < ntiles) stage_dma(tile + 1, buf ^ 1);
    } else {
      stage_regs(tile);
    }
    if (p.S_tail > 0 && kv0 + KV > p.S_main) {  // block-uniform: patch the time-token rows into the tile
      if constexpr (!L::DMA) __syncthreads();
      const int sample = seq % p.tail_mod;
      for (int e = tid; e < p.S_tail * DH; e += 256) {
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
        const typename P::Frag kf = P::load(&Ks[L::kidx(kt * 16 + l15, kc * P::KCH + g * P::EPL)]);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s[kt][qt] = P::mfma(kf, qf[qt][kc], s[kt][qt]);
      }
    }
    // ---- online softmax (log2 domain); lane owns query l15, keys kt*16 + g*4 + r ----
    // VALU diet (this loop is VALU-bound, not MFMA-bound): masking only on the last tile, the
    // 1/sqrt(dh)*log2(e) scale folded into the exp2 argument (one fma per score), and the O rescale
    // skipped (wave-uniform branch) unless some running max actually moved.
    if (kv0 + KV > S_total) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kv0 + kt * 16 + g * 4 + r >= S_total) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) s[kt][qt][r] = -INFINITY;
          }
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float mx = s[0][qt][0];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][qt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(mrun[qt], mx * p.scale_log2e);
      const float alpha = __builtin_amdgcn_exp2f(mrun[qt] - mnew);
      const bool moved = mnew > mrun[qt];
      mrun[qt] = mnew;
      float ps = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(fmaf(s[kt][qt][r], p.scale_log2e, -mnew));
          s[kt][qt][r] = e;
          ps += e;
        }
      lsum[qt] = lsum[qt] * alpha + ps;
      if (__any(moved)) {
#pragma unroll
        for (int dv = 0; dv < DVT; ++dv)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[qt][dv][r] *= alpha;
      }
    }
    // ---- O^T += V^T P^T ----
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {  // 32-key chunk: element e -> key c*32 + (e>>2)*16 + g*4 + (e&3)
        bf16x8 pf[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pf[qt][r] = (bf16_t)s[2 * c][qt][r];
            pf[qt][4 + r] = (bf16_t)s[2 * c + 1][qt][r];
          }
        }
#pragma unroll
        for (int dv = 0; dv < DVT; ++dv) {
          const int row = dv * 16 + l15;
          const bf16x4 lo = *reinterpret_cast<const bf16x4*>(&Vs[L::vidx(row, c * 32 + g * 4)]);
          const bf16x4 hi = *reinterpret_cast<const bf16x4*>(&Vs[L::vidx(row, c * 32 + 16 + g * 4)]);
          const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) o[qt][dv] = P::mfma(vf, pf[qt], o[qt][dv]);
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
  }

  // ---- normalise and store: lane owns query l15, rows dv*16 + g*4 + {0..3} ----
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
      if constexpr (sizeof(T) == 2) {
        bf16x4 ov = {(bf16_t)(v[0] * inv), (bf16_t)(v[1] * inv), (bf16_t)(v[2] * inv), (bf16_t)(v[3] * inv)};
        *reinterpret_cast<bf16x4*>(Op + dv * 16 + g * 4) = ov;
      } else {
        *reinterpret_cast<float4*>(Op + dv * 16 + g * 4) = make_float4(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv);
      }
    }
  }
}
