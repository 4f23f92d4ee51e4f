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

template <typename T, int DH>
struct AttnTile {
  static constexpr int QT = 2;             // 16-query tiles per wave
  static constexpr int BQ = 4 * QT * 16;   // queries per block (4 waves)
  static constexpr int KV = 64;            // keys per iteration
  static constexpr int PADK = sizeof(T) == 2 ? 8 : 2;
  static constexpr int PADV = sizeof(T) == 2 ? 4 : 2;
  static constexpr int LSK = DH + PADK;    // K tile row stride (elements)
  static constexpr int LSV = KV + PADV;    // V^T tile row stride
};

template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_kernel(AttnP p) {
  using P = Prec<T>;
  using A = AttnTile<T, DH>;
  constexpr int QT = A::QT, BQ = A::BQ, KV = A::KV, LSK = A::LSK, LSV = A::LSV;
  constexpr int KC = DH / P::KCH;   // k-chunks over head_dim (QK^T)
  constexpr int DVT = DH / 16;      // 16-row tiles of O^T
  constexpr int VEC = 16 / sizeof(T);
  __shared__ __attribute__((aligned(16))) T Ks[KV * LSK];
  __shared__ __attribute__((aligned(16))) T Vs[DH * LSV];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int seq = blockIdx.z, head = blockIdx.y;
  const int q0 = blockIdx.x * BQ + wid * (QT * 16);
  const int slot = p.kv_slot ? p.kv_slot[seq] : seq;
  const int S_total = p.S_main + p.S_tail;

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

  constexpr int KVEC_ROW = DH / VEC;           // 16B vectors per K row
  constexpr int KVECS = KV * KVEC_ROW;
  constexpr int VVEC_ROW = KV / VEC;           // 16B vectors per V^T row
  constexpr int VVECS = DH * VVEC_ROW;

  for (int kv0 = 0; kv0 < S_total; kv0 += KV) {
    __syncthreads();  // previous tile fully consumed
    for (int v = tid; v < KVECS; v += 256) {
      const int r = v / KVEC_ROW, c = v % KVEC_ROW;
      const uint4 val = *reinterpret_cast<const uint4*>(Kb + (int64_t)(kv0 + r) * p.ldk + c * VEC);
      if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<uint4*>(&Ks[r * LSK + c * VEC]) = val;
      } else {
        reinterpret_cast<uint2*>(&Ks[r * LSK + c * VEC])[0] = make_uint2(val.x, val.y);
        reinterpret_cast<uint2*>(&Ks[r * LSK + c * VEC])[1] = make_uint2(val.z, val.w);
      }
    }
    for (int v = tid; v < VVECS; v += 256) {
      const int r = v / VVEC_ROW, c = v % VVEC_ROW;
      const uint4 val = *reinterpret_cast<const uint4*>(Vb + (int64_t)r * p.ldvt + kv0 + c * VEC);
      reinterpret_cast<uint2*>(&Vs[r * LSV + c * VEC])[0] = make_uint2(val.x, val.y);
      reinterpret_cast<uint2*>(&Vs[r * LSV + c * VEC])[1] = make_uint2(val.z, val.w);
    }
    if (p.S_tail > 0 && kv0 + KV > p.S_main) {  // block-uniform: patch the time-token rows
      __syncthreads();
      const int sample = seq % p.tail_mod;
      for (int e = tid; e < p.S_tail * DH; e += 256) {
        const int j = e / DH, c = e % DH;
        const int kl = p.S_main + j - kv0;
        if (kl >= 0 && kl < KV) {
          const int64_t off = (int64_t)sample * p.tail_sample_stride + (int64_t)j * p.tail_row_stride + head * DH + c;
          Ks[kl * LSK + c] = from_f32<T>(p.ktail[off]);
          Vs[c * LSV + kl] = from_f32<T>(p.vtail[off]);
        }
      }
    }
    __syncthreads();

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
        const typename P::Frag kf = P::load(&Ks[(kt * 16 + l15) * LSK + kc * P::KCH + g * P::EPL]);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s[kt][qt] = P::mfma(kf, qf[qt][kc], s[kt][qt]);
      }
    }
    // ---- online softmax (log2 domain); lane owns query l15, keys kt*16 + g*4 + r ----
    const bool partial = kv0 + KV > S_total;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = s[kt][qt][r] * p.scale_log2e;
          if (partial && kv0 + kt * 16 + g * 4 + r >= S_total) v = -INFINITY;
          s[kt][qt][r] = v;
          mx = fmaxf(mx, v);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(mrun[qt], mx);
      const float alpha = exp2f(mrun[qt] - mnew);
      mrun[qt] = mnew;
      float ps = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = exp2f(s[kt][qt][r] - mnew);
          s[kt][qt][r] = e;
          ps += e;
        }
      lsum[qt] = lsum[qt] * alpha + ps;
#pragma unroll
      for (int dv = 0; dv < DVT; ++dv)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[qt][dv][r] *= alpha;
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
          const T* vp = &Vs[(dv * 16 + l15) * LSV + c * 32 + g * 4];
          const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vp);
          const bf16x4 hi = *reinterpret_cast<const bf16x4*>(vp + 16);
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
            const float vf = Vs[(dv * 16 + l15) * LSV + kt * 16 + g * 4 + r];
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
