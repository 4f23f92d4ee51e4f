// MFMA GEMM kernels for the denoiser:  C[M,N] = A[M,K] * W[N,K]^T (+bias) with fused epilogues.
//
//  * gemm_kernel<T>      : 128x128 block tile, 4 waves (2x2), 16x16 MFMA fragments, LDS-staged
//                          (register prefetch of the next k-tile), T = float (fp32 MFMA, parity mode)
//                          or bf16 (throughput mode).  Epilogues: bias+activation store, transposed
//                          store (V^T for attention), FiLM-affine + residual update of the fp32 stream
//                          (transformer_modules.py:122-124,193), dilated-conv tap accumulation with
//                          LeakyReLU/avg-skip (model/diffusion.py:214-224).
//  * skinny_gemm_kernel  : M <= 64 rows (the per-step time/FiLM path), fp32 MFMA, operands straight
//                          from global memory, split-K over the 4 waves of a block.
#pragma once
#include "a2p_common.h"

enum { EPI_STORE = 0, EPI_STORE_T = 1, EPI_FILM_RES = 2, EPI_CONV = 3 };

struct GemmP {
  const void* A;
  const void* W;
  const float* bias;
  void* out;
  int64_t lda, ldw, ldo;
  int M, N, K;
  int ntaps;
  int64_t a_tap_stride, w_tap_stride;  // elements
  int epi, act, out_f32;
  float* resid;  // fp32 residual stream, updated in place (EPI_FILM_RES)
  int64_t ldx;
  const float* film;  // scale at film[seq*film_seq_stride + n], shift at +film_shift_off; NULL = plain residual
  int64_t film_seq_stride;
  int film_shift_off;
  int rows_per_seq;
  int64_t t_seq_stride;  // EPI_STORE_T: out[(m / rows_per_seq) * t_seq_stride + n * ldo + m % rows_per_seq]
  int out_seq_pad;       // EPI_STORE / EPI_CONV: output row = m + (m / rows_per_seq) * out_seq_pad
  const void* skip;      // EPI_CONV: (skip[m*ld_skip + n] + y) / 2 when non-null (dtype T)
  int64_t ld_skip;
};

template <typename T>
struct GemmTile {
  static constexpr int BM = 128, BN = 128;
  static constexpr int BK = sizeof(T) == 2 ? 64 : 32;
  static constexpr int PAD = sizeof(T) == 2 ? 8 : 2;
  static constexpr int LS = BK + PAD;       // LDS row stride (elements)
  static constexpr int VEC = 16 / sizeof(T);
  static constexpr int VPR = BK / VEC;      // 16-byte vectors per tile row (= 8)
};

template <typename T>
__device__ __forceinline__ void lds_store16(T* dst, uint4 v) {
  if constexpr (sizeof(T) == 2) {
    *reinterpret_cast<uint4*>(dst) = v;
  } else {  // fp32 rows are only 8-byte aligned (stride 34 floats)
    reinterpret_cast<uint2*>(dst)[0] = make_uint2(v.x, v.y);
    reinterpret_cast<uint2*>(dst)[1] = make_uint2(v.z, v.w);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP p) {
  using P = Prec<T>;
  using G = GemmTile<T>;
  constexpr int BM = G::BM, BN = G::BN, BK = G::BK, LS = G::LS, VEC = G::VEC;
  __shared__ __attribute__((aligned(16))) T As[BM * LS];
  __shared__ __attribute__((aligned(16))) T Ws[BN * LS];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  const T* __restrict__ A = reinterpret_cast<const T*>(p.A);
  const T* __restrict__ W = reinterpret_cast<const T*>(p.W);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-thread staging coordinates: 4 vectors of A and 4 of W per k-tile
  int srow[4], scv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int v = tid + 256 * i;
    srow[i] = v >> 3;
    scv[i] = v & 7;
  }
  const int ktiles = p.K / BK;
  const int total = ktiles * p.ntaps;

  uint4 ra[4], rw[4];
  auto gload = [&](int it) {
    const int tap = it / ktiles;
    const int k0 = (it - tap * ktiles) * BK;
    const T* At = A + tap * p.a_tap_stride + k0;
    const T* Wt = W + tap * p.w_tap_stride + k0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gm = m0 + srow[i], gn = n0 + srow[i];
      ra[i] = gm < p.M ? *reinterpret_cast<const uint4*>(At + (int64_t)gm * p.lda + scv[i] * VEC) : make_uint4(0, 0, 0, 0);
      rw[i] = gn < p.N ? *reinterpret_cast<const uint4*>(Wt + (int64_t)gn * p.ldw + scv[i] * VEC) : make_uint4(0, 0, 0, 0);
    }
  };

  gload(0);
  for (int it = 0; it < total; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lds_store16<T>(&As[srow[i] * LS + scv[i] * VEC], ra[i]);
      lds_store16<T>(&Ws[srow[i] * LS + scv[i] * VEC], rw[i]);
    }
    __syncthreads();
    if (it + 1 < total) gload(it + 1);
#pragma unroll
    for (int kk = 0; kk < BK / P::KCH; ++kk) {
      typename P::Frag af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = P::load(&As[(wm * 64 + i * 16 + l15) * LS + kk * P::KCH + g * P::EPL]);
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = P::load(&Ws[(wn * 64 + j * 16 + l15) * LS + kk * P::KCH + g * P::EPL]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = P::mfma(wf[j], af[i], acc[i][j]);  // acc[r] = C[m=l15][n=g*4+r]
    }
    __syncthreads();
  }

  // ---- epilogue: lane owns row m = ..+l15 and 4 consecutive columns n = ..+g*4+{0..3} ----
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + l15;
    if (m >= p.M) continue;
    int seq = 0, sm = m;
    if (p.epi == EPI_FILM_RES || p.epi == EPI_STORE_T || p.out_seq_pad) {
      seq = m / p.rows_per_seq;
      sm = m - seq * p.rows_per_seq;
    }
    const int64_t orow = (int64_t)m + (int64_t)seq * p.out_seq_pad;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + g * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      }
      if (p.epi == EPI_FILM_RES) {
        float4* xp = reinterpret_cast<float4*>(p.resid + (int64_t)m * p.ldx + n);
        float4 x = *xp;
        if (p.film) {
          const float* fp = p.film + (int64_t)seq * p.film_seq_stride + n;
          const float4 sc = *reinterpret_cast<const float4*>(fp);
          const float4 sh = *reinterpret_cast<const float4*>(fp + p.film_shift_off);
          x.x += (sc.x + 1.0f) * v[0] + sh.x;
          x.y += (sc.y + 1.0f) * v[1] + sh.y;
          x.z += (sc.z + 1.0f) * v[2] + sh.z;
          x.w += (sc.w + 1.0f) * v[3] + sh.w;
        } else {
          x.x += v[0]; x.y += v[1]; x.z += v[2]; x.w += v[3];
        }
        *xp = x;
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act);
      if (p.epi == EPI_CONV && p.skip) {
        const T* sp = reinterpret_cast<const T*>(p.skip) + (int64_t)m * p.ld_skip + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (to_f32(sp[r]) + v[r]) * 0.5f;
      }
      if (p.epi == EPI_STORE_T) {
        T* op = reinterpret_cast<T*>(p.out) + (int64_t)seq * p.t_seq_stride + (int64_t)n * p.ldo + sm;
#pragma unroll
        for (int r = 0; r < 4; ++r) op[(int64_t)r * p.ldo] = from_f32<T>(v[r]);
      } else if (p.out_f32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + orow * p.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        T* op = reinterpret_cast<T*>(p.out) + orow * p.ldo + n;
        if constexpr (sizeof(T) == 2) {
          bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          *reinterpret_cast<bf16x4*>(op) = o;
        } else {
          *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Skinny fp32 GEMM for the per-step time path (a15, a20): out[M<=64, N] = act(A[M,K] W[N,K]^T + b).
// One block = one 16-column strip; its 4 waves split K; operands come straight from global
// memory as float4 (k index of MFMA j inside a 16-wide k-slab: 4*g + j for both operands).
// ---------------------------------------------------------------------------------------------
struct SkinnyP {
  const float* A;   // [M, K] fp32
  const float* W;   // [N, K] fp32
  const float* bias;
  float* out;       // [M, ldo]
  int64_t lda, ldw, ldo;
  int M, N, K;      // K % 64 == 0, N % 16 == 0
  int act;
};

__global__ __launch_bounds__(256) void skinny_gemm_kernel(SkinnyP p) {
  __shared__ float red[4][4][64][4];  // [wave][mtile][lane][reg]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int mt = (p.M + 15) >> 4;  // 1..4 row tiles
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int kper = p.K >> 2;  // per wave
  const float* Wr = p.W + (int64_t)(n0 + l15) * p.ldw + wid * kper + g * 4;
  const float* Ar[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = i * 16 + l15;
    if (m >= p.M) m = p.M - 1;  // clamped rows are never stored
    Ar[i] = p.A + (int64_t)m * p.lda + wid * kper + g * 4;
  }
  for (int k = 0; k < kper; k += 16) {
    const float4 w = *reinterpret_cast<const float4*>(Wr + k);
    const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < mt) {
        const float4 a = *reinterpret_cast<const float4*>(Ar[i] + k);
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[j], av[j], acc[i], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wid][i][lane][r] = acc[i][r];
  __syncthreads();
  // wave w finalises row tile w: acc[r] = C[m = l15][n = g*4 + r]
  if (wid < mt) {
    const int m = wid * 16 + l15;
    if (m < p.M) {
      const int n = n0 + g * 4;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = red[0][wid][lane][r] + red[1][wid][lane][r] + red[2][wid][lane][r] + red[3][wid][lane][r];
        if (p.bias) s += p.bias[n + r];
        v[r] = apply_act(s, p.act);
      }
      *reinterpret_cast<float4*>(p.out + (int64_t)m * p.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}
