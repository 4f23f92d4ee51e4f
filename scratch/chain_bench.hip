// Ablation bench of the chain kernels on synthetic buffers (scratch; not product).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/chain_bench.hip -o scratch/chain_bench
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../audio2photoreal_amd/csrc/kernels_chain.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <typename F>
float time_it(F f, int iters = 10) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) f();
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters * 1e3f;
}
static int g_thrash = 0;  // 1: read a 512 MiB buffer between launches (evicts L2 and the 256 MiB MALL like the K/V caches do)
static uint4* g_tbuf = nullptr;
static uint4* g_tout = nullptr;
__global__ void thrash_kernel(const uint4* __restrict__ p, size_t n_per_block, uint4* out) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  const uint4* q = p + (size_t)blockIdx.x * n_per_block;
  for (size_t i = threadIdx.x; i < n_per_block; i += blockDim.x) { uint4 v = q[i]; acc.x ^= v.x; acc.y ^= v.y; }
  if (acc.x == 0x12345678) out[0] = acc;
}
static int g_cycle = 1;  // number of distinct weight streams cycled through (1 = always L2-hot)
template <int MT, int MODE, int ABL, int NW = 4>
void run(const char* name, ChainP p, int stages) {
  const int grid = (p.M + 16 * MT - 1) / (16 * MT);
  const h16_t* base = p.stream;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double tot = 0; const int iters = 8;
  for (int it = 0; it < iters + 2; ++it) {
    p.stream = base + (size_t)(it % g_cycle) * (256 + 8) * 8192;
    if (g_thrash) thrash_kernel<<<2048, 256>>>(g_tbuf, ((size_t)512 << 20) / 16 / 2048, g_tout);
    hipExtLaunchKernelGGL((chain_kernel<512, MT, MODE, ABL, NW>), dim3(grid), dim3(64 * NW), 0, 0, e0, e1, 0, p);
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 2) tot += ms;
  }
  const float us = tot / iters * 1e3;
  printf("  NW=%d MT=%d mode=%d abl=%2d %-28s %8.1f us  (%d blocks, %.3f us/stage)\n", NW, MT, MODE, ABL, name, us, grid, us / stages / ((grid + 255) / 256));
}
template <int MT, int NW>
void all(ChainP p) {
  p.has_next = 1;
  run<MT, CHAIN_PRE, 0, NW>("full", p, 96);
  run<MT, CHAIN_PRE, 1, NW>("no stores", p, 96);
  run<MT, CHAIN_MID, 0, NW>("full", p, 64);
  run<MT, CHAIN_MID, 1, NW>("no stores", p, 64);
  run<MT, CHAIN_POST, 0, NW>("full", p, 256);
  run<MT, CHAIN_POST, 1, NW>("no stores", p, 256);
  run<MT, CHAIN_POST, 5, NW>("no stores, no dma", p, 256);
  {
    unsigned long long* st; CK(hipMalloc(&st, 64 * 8)); CK(hipMemset(st, 0, 64 * 8));
    ChainP q = p; q.fin_out = reinterpret_cast<float*>(st);
    run<MT, CHAIN_POST, 64, NW>("full + stamps", q, 256);
    unsigned long long h[64]; CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
    static const char* names[] = {"prologue", "out_proj", "film_res", "ln_stats", "ln_write", "ffn", "film_res", "store_x", "ln+rope", "qk_gemm", "ln+v_gemm", "drain"};
    for (int b = 0; b < 2; ++b) {
      printf("    block %3d phases (us):", b ? 101 : 0);
      for (int i = 1; i <= 12; ++i) printf(" %s=%.2f", names[i - 1], (double)(h[b * 32 + i] - h[b * 32 + i - 1]) * 0.01);
      printf("  total=%.2f\n", (double)(h[b * 32 + 12] - h[b * 32]) * 0.01);
    }
    CK(hipFree(st));
  }
}
int main(int argc, char** argv) {
  for (int M : {9600}) {
    const int D = 512;
    float *x, *aux, *vec, *film; h16_t *ain, *stream, *qk, *vt; float2* cs;
    CK(hipMalloc(&x, (size_t)M * D * 4)); CK(hipMalloc(&ain, (size_t)M * D * 2)); CK(hipMalloc(&stream, (size_t)(16 * (256 + 8) + 64) * 16384));
    CK(hipMalloc(&aux, 16384)); CK(hipMalloc(&vec, 8192 * 4)); CK(hipMalloc(&film, (size_t)64 * 4 * D * 4));
    CK(hipMalloc(&qk, (size_t)M * 2 * D * 2)); CK(hipMalloc(&vt, (size_t)M * D * 2 + (1 << 20))); CK(hipMalloc(&cs, (size_t)640 * 256 * 8));
    std::vector<uint16_t> h((size_t)(16 * (256 + 8) + 64) * 8192);
    for (auto& v : h) v = 0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15);
    CK(hipMemcpy(stream, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    h.resize((size_t)M * D);
    CK(hipMemcpy(ain, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(x, 0, (size_t)M * D * 4)); CK(hipMemset(aux, 0, 16384)); CK(hipMemset(vec, 0, 8192 * 4)); CK(hipMemset(film, 0, (size_t)64 * 4 * D * 4));
    CK(hipMemset(cs, 0, (size_t)640 * 256 * 8));
    ChainP p; memset(&p, 0, sizeof(p));
    p.M = M; p.rows_per_seq = 600; p.aux_kb = 10; p.x = x; p.stream = stream; p.aux = aux; p.ain = ain; p.ld_ain = D;
    p.bias_o = vec; p.film_o = film; p.film_seq_stride = 4 * D; p.film_shift_off = D; p.lnA_g = vec + 512; p.lnA_b = vec + 1024;
    p.q_out = qk; p.ld_q = D; p.bias_2 = vec + 1536; p.film_f = film + 2 * D; p.lnB_g = vec + 2048; p.lnB_b = vec + 2560;
    p.qk_out = qk; p.ld_qk = 2 * D; p.vt_out = vt; p.vt_seq_stride = (int64_t)D * 640; p.ld_vt = 640; p.cst = reinterpret_cast<const f32x4*>(cs); p.cs_npos = 640;
    CK(hipMalloc(&g_tbuf, (size_t)512 << 20)); CK(hipMalloc(&g_tout, 64)); CK(hipMemset(g_tbuf, 1, (size_t)512 << 20));
    for (int th : {1, 0}) {
      g_cycle = 16; g_thrash = th;
      printf("M=%d weight streams cycled=16, MALL/L2 thrash between launches=%d\n", M, th);
      all<3, 4>(p);
      all<3, 8>(p);
      if (getenv("CB_MT4")) { all<4, 4>(p); all<4, 8>(p); }

    }
  }
  return 0;
}
