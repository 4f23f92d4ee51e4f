// Bench of the generation-3 chain kernels (two workgroups per CU) next to generation 1 on synthetic buffers with in-kernel phase stamps (scratch; not product).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DA2P_HALF -DC3_STAMPS scratch/chain3_bench.hip -o scratch/chain3_bench
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "kernels_chain3.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
static uint4* g_tbuf = nullptr;
static uint4* g_tout = nullptr;
static int g_thrash = 1, g_cycle = 1, g_phase = 0, g_xpf = 0;
__global__ void thrash_kernel(const uint4* __restrict__ p, size_t n_per_block, uint4* out) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  const uint4* q = p + (size_t)blockIdx.x * n_per_block;
  for (size_t i = threadIdx.x; i < n_per_block; i += blockDim.x) { uint4 v = q[i]; acc.x ^= v.x; acc.y ^= v.y; }
  if (acc.x == 0x12345678) out[0] = acc;
}
template <int GEN, int MT, int MODE>
void run(ChainP p0, int stages, unsigned long long* st) {
  Chain3P p;
  static_cast<ChainP&>(p) = p0;
  const int grid = (p.M + 16 * MT - 1) / (16 * MT);
  const h16_t* base = p.stream;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double tot = 0; const int iters = 6;
  for (int it = 0; it < iters + 2; ++it) {
    p.stream = base + (size_t)(it % 8) * (256 + 8) * 8192;
    if (g_thrash) thrash_kernel<<<2048, 256>>>(g_tbuf, ((size_t)512 << 20) / 16 / 2048, g_tout);
    if (!g_cycle) p.stream = base;
    p.n_pf = 0; p.n_stages = stages; p.phase_us = g_phase; p.phase_blocks = 512; p.x_prefetch = g_xpf;
    if constexpr (GEN == 3) hipExtLaunchKernelGGL((chain3_kernel<512, MT, MODE>), dim3(grid), dim3(256), 0, 0, e0, e1, 0, p);
    else if constexpr (GEN == 4) hipExtLaunchKernelGGL((chain_kernel<512, MT, MODE, 0, 4>), dim3(grid), dim3(256), 0, 0, e0, e1, 0, p);
    else hipExtLaunchKernelGGL((chain_kernel<512, MT, MODE, 0, 8>), dim3(grid), dim3(512), 0, 0, e0, e1, 0, p);
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 2) tot += ms;
  }
  const float us = tot / iters * 1e3;
  const int rounds = (grid + (GEN == 3 ? 511 : 255)) / (GEN == 3 ? 512 : 256);
  printf("  gen %d rows %2d mode %d: %8.1f us  (%3d blocks = %d rounds, %.3f us/stage/round, %.2f ns per row-stage)\n", GEN, 16 * MT, MODE, us, grid, rounds,
         us / stages / rounds, 1e3 * us / stages / rounds / (16 * MT));
  if (GEN == 3 && MODE == CHAIN_POST) {
    unsigned long long h[64]; CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
    static const char* names[] = {"prologue", "out_proj", "film_res", "ln+park", "ffn", "film_res", "ln+rope+store", "-", "-", "qk_gemm", "reload+ln", "v_gemm"};
    static const int idx[] = {1, 2, 3, 4, 5, 6, 9, 10, 11, 12};
    for (int b = 0; b < 2; ++b) {
      printf("      block %3d phases (us):", b ? 301 : 0);
      int prev = 0;
      for (int i : idx) { printf(" %s=%.2f", names[i - 1], (double)(h[b * 32 + i] - h[b * 32 + prev]) * 0.01); prev = i; }
      printf("  total=%.2f\n", (double)(h[b * 32 + 13] - h[b * 32]) * 0.01);
    }
  }
}
int main(int argc, char** argv) {
  if (argc > 1) g_phase = atoi(argv[1]);
  if (argc > 2) g_xpf = atoi(argv[2]);
  printf("phase_us=%d x_prefetch=%d\n", g_phase, g_xpf);
  const int D = 512, Mmax = 38400;
  float *x, *aux, *vec, *film; h16_t *ain, *stream, *qk, *vt; float2* cs;
  CK(hipMalloc(&x, (size_t)Mmax * D * 4)); CK(hipMalloc(&ain, (size_t)Mmax * D * 2)); CK(hipMalloc(&stream, (size_t)(8 * (256 + 8) + 64) * 16384));
  CK(hipMalloc(&aux, 16384)); CK(hipMalloc(&vec, 8192 * 4)); CK(hipMalloc(&film, (size_t)64 * 4 * D * 4));
  CK(hipMalloc(&qk, (size_t)Mmax * 2 * D * 2)); CK(hipMalloc(&vt, (size_t)Mmax * D * 2 + (1 << 20))); CK(hipMalloc(&cs, (size_t)640 * 256 * 8));
  std::vector<uint16_t> h((size_t)(8 * (256 + 8) + 64) * 8192);
  for (auto& v : h) v = 0x2c00 + (rand() & 0x3ff) - ((rand() & 1) << 15);
  CK(hipMemcpy(stream, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  h.resize((size_t)Mmax * D);
  CK(hipMemcpy(ain, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(x, 0, (size_t)Mmax * D * 4)); CK(hipMemset(aux, 0, 16384)); CK(hipMemset(vec, 0, 8192 * 4)); CK(hipMemset(film, 0, (size_t)64 * 4 * D * 4));
  CK(hipMemset(cs, 0, (size_t)640 * 256 * 8));
  unsigned long long* st; CK(hipMalloc(&st, 64 * 8)); CK(hipMemset(st, 0, 64 * 8));
  CK(hipMalloc(&g_tbuf, (size_t)512 << 20)); CK(hipMalloc(&g_tout, 64)); CK(hipMemset(g_tbuf, 1, (size_t)512 << 20));
  for (int mode = 0; mode < 2; ++mode)
  for (int M : {9600, 38400}) {
    g_thrash = mode == 0; g_cycle = mode == 0;
    if (mode == 1 && M != 9600) continue;
    ChainP p; memset(&p, 0, sizeof(p));
    p.M = M; p.rows_per_seq = 600; p.aux_kb = 10; p.x = x; p.stream = stream; p.aux = aux; p.ain = ain; p.ld_ain = D;
    p.bias_o = vec; p.film_o = film; p.film_seq_stride = 4 * D; p.film_shift_off = D; p.lnA_g = vec + 512; p.lnA_b = vec + 1024;
    p.q_out = qk; p.ld_q = D; p.bias_2 = vec + 1536; p.film_f = film + 2 * D; p.lnB_g = vec + 2048; p.lnB_b = vec + 2560;
    p.qk_out = qk; p.ld_qk = 2 * D; p.vt_out = vt; p.vt_seq_stride = (int64_t)D * 640; p.ld_vt = 640; p.cst = reinterpret_cast<const f32x4*>(cs); p.cs_npos = 640;
    p.has_next = 1; p.x_in_tiled = 1; p.x_out_tiled = 1; p.fin_out = reinterpret_cast<float*>(st);
    printf("M=%d (%s)\n", M, g_thrash ? "L2/MALL thrashed between launches, 8 streams cycled" : "WARM: same stream every launch, no thrash");
    run<1, 3, CHAIN_POST>(p, 256, st);
    run<4, 3, CHAIN_POST>(p, 256, st);
    run<3, 3, CHAIN_POST>(p, 256, st);
    run<3, 2, CHAIN_POST>(p, 256, st);
    run<1, 3, CHAIN_MID>(p, 64, st);
    run<3, 3, CHAIN_MID>(p, 64, st);
    p.x_in_tiled = 0;
    run<1, 3, CHAIN_PRE>(p, 96, st);
    run<3, 3, CHAIN_PRE>(p, 96, st);
    p.x_in_tiled = 1;
  }
  return 0;
}
