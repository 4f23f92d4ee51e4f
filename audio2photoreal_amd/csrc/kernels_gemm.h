// MFMA GEMM kernels for the denoiser:  C[M,N] = A[M,K] * W[N,K]^T (+bias) with fused epilogues.
//
//  * gemm_kernel<T, MT, EPI, ACT, OUTF32>
//        (32*MT)x128 block tile (MT = 4: 128x128, MT = 2: 64x128 for launches that would not fill
//        256 CUs), 4 waves as 2x2, 16x16 MFMA fragments, K-step 64 (bf16) / 32 (fp32).
//        bf16 (throughput mode): operands go HBM -> LDS directly (global_load_lds, 16 B per lane),
//        double-buffered, XOR-swizzled on the SOURCE address + on the fragment read so ds_read_b128 is
//        bank-conflict free with unpadded 128-byte rows (cdna_hip_programming.md §5.4 rule 21, T2).
//        fp32 (parity mode, v_mfma_f32_16x16x4_f32): register-staged single LDS buffer.
//        Epilogues are compile-time: bias+activation store (T or fp32), transposed store (V^T for
//        attention), FiLM-affine + residual update of the fp32 stream (transformer_modules.py:122-124,
//        193), dilated-conv tap accumulation with LeakyReLU / averaged skip (model/diffusion.py:214-224).
//  * skinny_gemm_kernel : M <= 64 rows (per-step time / FiLM path), fp32 MFMA, operands straight from
//        global memory, split-K over the 4 waves of a block.
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
  // split-operand rows (kernels_misc.h split3_kernel): split_third > 0 -> the 16-bit output row is [hi | lo | hi] with the pieces
  // split_third elements apart (the A' operand of the next split GEMM); skip_lo > 0 -> the skip operand is such a row too and its
  // value is skip[n] + skip[skip_lo + n]
  int split_third, skip_lo;
  int64_t dup_off;   // fp32 EPI_STORE: != 0 -> every value is stored a second time dup_off elements further (both guidance halves of the
                     // input projection in one launch instead of a copy kernel behind it)
};

template <int ACT, bool FAST>
__device__ __forceinline__ float act_ct(float x) {
  if constexpr (ACT == ACT_GELU) return FAST ? act_gelu_fast(x) : act_gelu(x);
  else if constexpr (ACT == ACT_MISH) return act_mish(x);
  else if constexpr (ACT == ACT_SILU) return act_silu(x);
  else if constexpr (ACT == ACT_LRELU) return act_lrelu02(x);
  else if constexpr (ACT == ACT_RELU) return x > 0.0f ? x : 0.0f;
  else return x;
}

// ---- epilogue: lane owns row m = m_base + i*16 + l15 and 4 consecutive columns n = n_base + j*16 + g*4 ----
template <typename T, int MT, int EPI, int ACT, bool OUTF32>
__device__ __forceinline__ void gemm_epilogue(const GemmP& p, f32x4 (&acc)[MT][4], int m_base, int n_base, int l15, int g) {
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m_base + i * 16 + l15;
    if (m >= p.M) continue;
    int seq = 0, sm = m;
    if (EPI == EPI_FILM_RES || EPI == EPI_STORE_T || p.out_seq_pad) {
      seq = m / p.rows_per_seq;
      sm = m - seq * p.rows_per_seq;
    }
    const int64_t orow = (int64_t)m + (int64_t)seq * p.out_seq_pad;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n_base + j * 16 + g * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      }
      if constexpr (EPI == EPI_FILM_RES) {
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
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = act_ct<ACT, sizeof(T) == 2>(v[r]);
        if constexpr (EPI == EPI_CONV) {
          if (p.skip) {
            const T* sp = reinterpret_cast<const T*>(p.skip) + (int64_t)m * p.ld_skip + n;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = ((p.skip_lo ? to_f32(sp[r]) + to_f32(sp[p.skip_lo + r]) : to_f32(sp[r])) + v[r]) * 0.5f;
          }
        }
        if constexpr (EPI == EPI_STORE_T) {
          T* op = reinterpret_cast<T*>(p.out) + (int64_t)seq * p.t_seq_stride + (int64_t)n * p.ldo + sm;
#pragma unroll
          for (int r = 0; r < 4; ++r) op[(int64_t)r * p.ldo] = from_f32<T>(v[r]);
        } else if constexpr (OUTF32) {
          float* op = reinterpret_cast<float*>(p.out) + orow * p.ldo + n;
          *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
          if (p.dup_off) *reinterpret_cast<float4*>(op + p.dup_off) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          T* op = reinterpret_cast<T*>(p.out) + orow * p.ldo + n;
          if constexpr (sizeof(T) == 2) {
            h16x4 o = {(h16_t)v[0], (h16_t)v[1], (h16_t)v[2], (h16_t)v[3]};
            *reinterpret_cast<h16x4*>(op) = o;
            if (p.split_third) {
              h16x4 lo;
#pragma unroll
              for (int r = 0; r < 4; ++r) lo[r] = (h16_t)(v[r] - (float)o[r]);
              *reinterpret_cast<h16x4*>(op + p.split_third) = lo;
              *reinterpret_cast<h16x4*>(op + 2 * p.split_third) = o;
            }
          } else {
            *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
    }
  }
}

// ---- fp32 main loop: register-staged, single LDS buffer, padded rows --------------------------------
template <int MT>
__device__ __forceinline__ void gemm_mainloop_f32(const GemmP& p, f32x4 (&acc)[MT][4], int m0, int n0) {
  using P = Prec<float>;
  constexpr int BM = 32 * MT, BN = 128, BK = 32, LS = BK + 2, VEC = 4;
  constexpr int AV = BM * 8 / 256;  // 16-byte vectors of A per thread (8 per row)
  __shared__ __attribute__((aligned(16))) float As[BM * LS];
  __shared__ __attribute__((aligned(16))) float Ws[BN * LS];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  const float* __restrict__ A = reinterpret_cast<const float*>(p.A);
  const float* __restrict__ W = reinterpret_cast<const float*>(p.W);
  const int ktiles = p.K / BK;
  uint4 ra[AV], rw[4];
  auto gload = [&](const float* At, const float* Wt) {
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      const int v = tid + 256 * i, row = v >> 3, cv = v & 7;
      int gm = m0 + row;
      gm = gm < p.M ? gm : p.M - 1;  // clamped rows are never stored
      ra[i] = *reinterpret_cast<const uint4*>(At + (int64_t)gm * p.lda + cv * VEC);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int v = tid + 256 * i, row = v >> 3, cv = v & 7;
      int gn = n0 + row;
      gn = gn < p.N ? gn : p.N - 1;
      rw[i] = *reinterpret_cast<const uint4*>(Wt + (int64_t)gn * p.ldw + cv * VEC);
    }
  };
  for (int tap = 0; tap < p.ntaps; ++tap) {
    const float* At = A + tap * p.a_tap_stride;
    const float* Wt = W + tap * p.w_tap_stride;
    gload(At, Wt);
    for (int kt = 0; kt < ktiles; ++kt) {
#pragma unroll
      for (int i = 0; i < AV; ++i) {
        const int v = tid + 256 * i, row = v >> 3, cv = v & 7;
        reinterpret_cast<uint2*>(&As[row * LS + cv * VEC])[0] = make_uint2(ra[i].x, ra[i].y);
        reinterpret_cast<uint2*>(&As[row * LS + cv * VEC])[1] = make_uint2(ra[i].z, ra[i].w);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int v = tid + 256 * i, row = v >> 3, cv = v & 7;
        reinterpret_cast<uint2*>(&Ws[row * LS + cv * VEC])[0] = make_uint2(rw[i].x, rw[i].y);
        reinterpret_cast<uint2*>(&Ws[row * LS + cv * VEC])[1] = make_uint2(rw[i].z, rw[i].w);
      }
      __syncthreads();
      if (kt + 1 < ktiles) gload(At + (kt + 1) * BK, Wt + (kt + 1) * BK);
#pragma unroll
      for (int kk = 0; kk < BK / P::KCH; ++kk) {
        float af[MT], wf[4];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = As[(wm * 16 * MT + i * 16 + l15) * LS + kk * P::KCH + g];
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = Ws[(wn * 64 + j * 16 + l15) * LS + kk * P::KCH + g];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = P::mfma(wf[j], af[i], acc[i][j]);  // acc[r] = C[m=l15][n=g*4+r]
      }
      __syncthreads();
    }
  }
}

// ---- bf16 main loop: global_load_lds (16 B / lane), 2 LDS buffers, swizzled unpadded rows -----------
// LDS tile = [rows][64 bf16] (128-byte rows = 8 chunks of 16 B).  Chunk c of row r lives at chunk
// position c ^ ((r >> 1) & 7): conflict-free for the 16-lane groups of ds_read_b128.
__device__ __forceinline__ void glds16(const h16_t* gsrc, h16_t* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// NB: depth of the tile ring.  2 = the next k-tile is requested while the current one is consumed (several workgroups per CU hide
// each other's waits: the large launches).  4 = three k-tiles in flight with a counted vmcnt wait: for launches of at most one
// workgroup per CU -- the 480-row forwards of BASELINE configs[0] -- whose time is K/64 serial HBM/L2 round trips otherwise.
template <int MT, int NB = 2>
__device__ __forceinline__ void gemm_mainloop_bf16(const GemmP& p, f32x4 (&acc)[MT][4], int m0, int n0) {
  using P = Prec<h16_t>;
  constexpr int BM = 32 * MT, BN = 128, BK = 64;
  constexpr int AI = BM / 32;  // glds instructions per wave for the A tile (8 rows each, 4 waves)
  constexpr int PCS = AI + 4;  // DMA instructions per wave per k-tile
  __shared__ __attribute__((aligned(16))) h16_t smem[NB * (BM + BN) * BK];
  h16_t* const As0 = smem;
  h16_t* const Ws0 = smem + NB * BM * BK;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  const h16_t* __restrict__ A = reinterpret_cast<const h16_t*>(p.A);
  const h16_t* __restrict__ W = reinterpret_cast<const h16_t*>(p.W);
  const int ktiles = p.K / BK;
  const int total = ktiles * p.ntaps;

  // per-lane source offsets (elements) of the staging loads: row = grp*8 + lane/8, LDS chunk position lane%8
  const int lrow = lane >> 3, lpos = lane & 7;
  int64_t a_off[AI], w_off[4];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int row = (i * 4 + wid) * 8 + lrow;
    int gm = m0 + row;
    gm = gm < p.M ? gm : p.M - 1;  // clamped rows are never stored
    a_off[i] = (int64_t)gm * p.lda + ((lpos ^ ((row >> 1) & 7)) << 3);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 4 + wid) * 8 + lrow;
    int gn = n0 + row;
    gn = gn < p.N ? gn : p.N - 1;
    w_off[i] = (int64_t)gn * p.ldw + ((lpos ^ ((row >> 1) & 7)) << 3);
  }
  auto stage = [&](int it, int buf) {
    const int tap = it / ktiles;  // ntaps == 1 almost always: the division is off the critical path (loads in flight)
    const int k0 = (it - tap * ktiles) * BK;
    const h16_t* At = A + tap * p.a_tap_stride + k0;
    const h16_t* Wt = W + tap * p.w_tap_stride + k0;
    h16_t* Ab = As0 + buf * BM * BK;
    h16_t* Wb = Ws0 + buf * BN * BK;
#pragma unroll
    for (int i = 0; i < AI; ++i) glds16(At + a_off[i], Ab + (i * 4 + wid) * 8 * BK);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(Wt + w_off[i], Wb + (i * 4 + wid) * 8 * BK);
  };
  // fragment read offsets (elements) inside a tile: row R, chunk (kk*4 + g) ^ ((R >> 1) & 7)
  int a_row[MT], w_row[4];
#pragma unroll
  for (int i = 0; i < MT; ++i) a_row[i] = wm * 16 * MT + i * 16 + l15;
#pragma unroll
  for (int j = 0; j < 4; ++j) w_row[j] = wn * 64 + j * 16 + l15;

#pragma unroll
  for (int i = 0; i < NB - 1; ++i)
    if (i < total) stage(i, i);
  int buf = 0, nbuf = NB - 1;   // ring slots of tile `it` and of tile it + NB - 1
  for (int it = 0; it < total; ++it) {
    if constexpr (NB == 2) {
      __syncthreads();  // drains vmcnt(0): tile `it` has landed; everyone finished reading buffer buf^1
    } else {
      // loads return in issue order: this wave's pieces of tile `it` are in once at most the pieces of the tiles requested after
      // it are outstanding (exactly min(NB - 2, total - 1 - it) tiles); the barrier publishes every wave's pieces and frees the
      // slot of tile it - 1 for the request below
      const int rem = total - 1 - it;
      if (rem >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * PCS) : "memory");
      else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PCS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      static_assert(NB == 2 || NB == 4, "ring depths 2 and 4");
    }
    if (it + NB - 1 < total) stage(it + NB - 1, nbuf);
    const h16_t* Ab = As0 + buf * BM * BK;
    const h16_t* Wb = Ws0 + buf * BN * BK;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      h16x8 af[MT], wf[4];
#pragma unroll
      for (int i = 0; i < MT; ++i)
        af[i] = *reinterpret_cast<const h16x8*>(Ab + a_row[i] * BK + (((kk * 4 + g) ^ ((a_row[i] >> 1) & 7)) << 3));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        wf[j] = *reinterpret_cast<const h16x8*>(Wb + w_row[j] * BK + (((kk * 4 + g) ^ ((w_row[j] >> 1) & 7)) << 3));
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = P::mfma(wf[j], af[i], acc[i][j]);
    }
    buf = buf + 1 == NB ? 0 : buf + 1;
    nbuf = nbuf + 1 == NB ? 0 : nbuf + 1;
  }
}

template <typename T, int MT, int EPI, int ACT, bool OUTF32, int NB = 2>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP p) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int m0 = blockIdx.y * (32 * MT), n0 = blockIdx.x * 128;
  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (sizeof(T) == 2) gemm_mainloop_bf16<MT, NB>(p, acc, m0, n0);
  else gemm_mainloop_f32<MT>(p, acc, m0, n0);
  gemm_epilogue<T, MT, EPI, ACT, OUTF32>(p, acc, m0 + (wid >> 1) * 16 * MT, n0 + (wid & 1) * 64, lane & 15, lane >> 4);
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

__device__ __forceinline__ void skinny_gemm_body(const SkinnyP& p, int block) {
  __shared__ float red[4][4][64][4];  // [wave][mtile][lane][reg]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int n0 = block * 16;
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
  // (unrolled: a wave's K slice is a chain of dependent 16-byte loads otherwise -- 32 round trips for time_mlp's K = 2048)
#pragma unroll 4
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

__global__ __launch_bounds__(256) void skinny_gemm_kernel(SkinnyP p) { skinny_gemm_body(p, blockIdx.x); }

// Up to three independent skinny GEMMs as ONE launch (the FiLM generators and the time tokens' K / V rows of the per-step time path
// all read tpath_post_kernel's outputs and nothing of each other): block ranges [0, nb0), [nb0, nb1), [nb1, grid).  A launch is
// ~4.6 us of dispatch floor on this part whatever it computes; same arithmetic per output element as three launches.
struct SkinnyG3 {
  SkinnyP p[3];
  int nb0, nb1;
};
__global__ __launch_bounds__(256) void skinny_gemm_group_kernel(SkinnyG3 g) {
  const int b = blockIdx.x;
  if (b < g.nb0) skinny_gemm_body(g.p[0], b);
  else if (b < g.nb1) skinny_gemm_body(g.p[1], b - g.nb0);
  else skinny_gemm_body(g.p[2], b - g.nb1);
}
