// Standalone ablation bench of the v1 GEMM kernel structure (scratch; not part of the product).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/gemm_bench.hip -o scratch/gemm_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../audio2photoreal_amd/csrc/kernels_gemm.h"

template <typename T>
struct GemmTile {
  static constexpr int BM = 128, BN = 128;
  static constexpr int BK = sizeof(T) == 2 ? 64 : 32;
  static constexpr int PAD = sizeof(T) == 2 ? 8 : 2;
  static constexpr int LS = BK + PAD;
  static constexpr int VEC = 16 / sizeof(T);
};
template <typename T>
__device__ __forceinline__ void lds_store16(T* dst, uint4 v) { *reinterpret_cast<uint4*>(dst) = v; }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// VARIANT bits: 1 = no epilogue stores, 2 = no global loads in loop, 4 = no MFMA, 8 = no LDS traffic in loop
template <int VARIANT>
__global__ __launch_bounds__(256) void gemm_abl(GemmP p) {
  using T = h16_t;
  using P = Prec<T>;
  using G = GemmTile<T>;
  constexpr int BM = G::BM, BN = G::BN, BK = G::BK, LS = G::LS, VEC = G::VEC;
  __shared__ __attribute__((aligned(16))) T As[BM * LS];
  __shared__ __attribute__((aligned(16))) T Ws[BN * LS];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const T* __restrict__ A = reinterpret_cast<const T*>(p.A);
  const T* __restrict__ W = reinterpret_cast<const T*>(p.W);
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int srow[4], scv[4];
  for (int i = 0; i < 4; ++i) { int v = tid + 256 * i; srow[i] = v >> 3; scv[i] = v & 7; }
  const int ktiles = p.K / BK;
  uint4 ra[4], rw[4];
  auto gload = [&](int it) {
    const int k0 = it * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gm = m0 + srow[i], gn = n0 + srow[i];
      ra[i] = gm < p.M ? *reinterpret_cast<const uint4*>(A + (int64_t)gm * p.lda + k0 + scv[i] * VEC) : make_uint4(0, 0, 0, 0);
      rw[i] = gn < p.N ? *reinterpret_cast<const uint4*>(W + (int64_t)gn * p.ldw + k0 + scv[i] * VEC) : make_uint4(0, 0, 0, 0);
    }
  };
  gload(0);
  for (int it = 0; it < ktiles; ++it) {
    if (!(VARIANT & 8) || it == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        lds_store16<T>(&As[srow[i] * LS + scv[i] * VEC], ra[i]);
        lds_store16<T>(&Ws[srow[i] * LS + scv[i] * VEC], rw[i]);
      }
    }
    __syncthreads();
    if (!(VARIANT & 2) && it + 1 < ktiles) gload(it + 1);
#pragma unroll
    for (int kk = 0; kk < BK / P::KCH; ++kk) {
      typename P::Frag af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = P::load(&As[(wm * 64 + i * 16 + l15) * LS + kk * P::KCH + g * P::EPL]);
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = P::load(&Ws[(wn * 64 + j * 16 + l15) * LS + kk * P::KCH + g * P::EPL]);
      if (!(VARIANT & 4)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = P::mfma(wf[j], af[i], acc[i][j]);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { asm volatile("" ::"v"(af[i])); asm volatile("" ::"v"(wf[i])); }
      }
    }
    __syncthreads();
  }
  if (VARIANT & 1) {
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + l15;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + g * 4;
      if (n >= p.N) continue;
      f32x4 v = acc[i][j];
      T* op = reinterpret_cast<T*>(p.out) + (int64_t)m * p.ldo + n;
      h16x4 o = {(h16_t)v[0], (h16_t)v[1], (h16_t)v[2], (h16_t)v[3]};
      *reinterpret_cast<h16x4*>(op) = o;
    }
  }
}

template <typename F>
float time_it(F f, int iters = 20) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters * 1e3f;  // us
}

int main() {
  const int shapes[][3] = {{9600, 512, 512}, {9600, 1024, 512}, {9600, 512, 1024}, {38400, 512, 512}, {38400, 1024, 512}};
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2];
    h16_t *A, *W, *O; float *X, *bias, *film;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&O, (size_t)M * N * 2 * 2));
    CK(hipMalloc(&X, (size_t)M * N * 4)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&film, (size_t)64 * 2 * N * 4));
    std::vector<uint16_t> h((size_t)M * K);
    for (auto& v : h) v = 0x3c00 + (rand() & 0x3ff);  // random-ish bf16 in [~0.008, ~0.03]
    CK(hipMemcpy(A, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice));
    h.resize((size_t)N * K);
    CK(hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice));
    CK(hipMemset(X, 0, (size_t)M * N * 4)); CK(hipMemset(bias, 0, N * 4)); CK(hipMemset(film, 0, (size_t)64 * 2 * N * 4));
    GemmP p; memset(&p, 0, sizeof(p));
    p.A = A; p.lda = K; p.W = W; p.ldw = K; p.out = O; p.ldo = N; p.M = M; p.N = N; p.K = K; p.ntaps = 1; p.rows_per_seq = 600;
    p.bias = bias;
    dim3 grid((N + 127) / 128, (M + 127) / 128);
    const double gf = 2.0 * M * N * K * 1e-9;
    auto rep = [&](const char* name, float us) { printf("  %-34s %8.1f us  %7.1f TF\n", name, us, gf / us * 1e-3); };
    printf("M=%d N=%d K=%d (%.1f GF, %d blocks)\n", M, N, K, gf, grid.x * grid.y);
    dim3 g4((N + 127) / 128, (M + 127) / 128), g2((N + 127) / 128, (M + 63) / 64);
    p.epi = EPI_STORE;
    rep("v2 MT4 STORE", time_it([&] { gemm_kernel<h16_t, 4, EPI_STORE, ACT_NONE, false><<<g4, 256>>>(p); }));
    rep("v2 MT2 STORE", time_it([&] { gemm_kernel<h16_t, 2, EPI_STORE, ACT_NONE, false><<<g2, 256>>>(p); }));
    rep("v2 MT4 STORE+GELU", time_it([&] { gemm_kernel<h16_t, 4, EPI_STORE, ACT_GELU, false><<<g4, 256>>>(p); }));
    rep("v2 MT2 STORE+GELU", time_it([&] { gemm_kernel<h16_t, 2, EPI_STORE, ACT_GELU, false><<<g2, 256>>>(p); }));
    p.resid = X; p.ldx = N; p.film = film; p.film_seq_stride = 2 * N; p.film_shift_off = N; p.epi = EPI_FILM_RES;
    rep("v2 MT4 FILM_RES", time_it([&] { gemm_kernel<h16_t, 4, EPI_FILM_RES, ACT_NONE, false><<<g4, 256>>>(p); }));
    rep("v2 MT2 FILM_RES", time_it([&] { gemm_kernel<h16_t, 2, EPI_FILM_RES, ACT_NONE, false><<<g2, 256>>>(p); }));
    p.epi = EPI_STORE_T; p.t_seq_stride = (int64_t)N * 640; p.ldo = 640;
    rep("v2 MT2 STORE_T", time_it([&] { gemm_kernel<h16_t, 2, EPI_STORE_T, ACT_NONE, false><<<g2, 256>>>(p); }));
    p.epi = EPI_STORE; p.ldo = N;
    rep("abl full", time_it([&] { gemm_abl<0><<<grid, 256>>>(p); }));
    rep("abl no-epilogue", time_it([&] { gemm_abl<1><<<grid, 256>>>(p); }));
    rep("abl no-gload", time_it([&] { gemm_abl<2><<<grid, 256>>>(p); }));
    rep("abl no-gload no-epi", time_it([&] { gemm_abl<3><<<grid, 256>>>(p); }));
    rep("abl no-mfma", time_it([&] { gemm_abl<4><<<grid, 256>>>(p); }));
    rep("abl no-gload no-epi no-ldswrite", time_it([&] { gemm_abl<11><<<grid, 256>>>(p); }));
    CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(O)); CK(hipFree(X)); CK(hipFree(bias)); CK(hipFree(film));
  }
  return 0;
}
