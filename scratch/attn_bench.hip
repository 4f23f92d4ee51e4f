// Ablation bench of the attention kernel (scratch; not product).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../audio2photoreal_amd/csrc/kernels_attn.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <typename F>
float time_it(F f, int iters = 20) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(e0)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters * 1e3f;
}
template <int ABL> void run(const char* name, AttnP p, int nseq, double gf) {
  p.nq = (p.Tq + 127) / 128; p.nheads = 8; p.nseq = nseq; p.xcd_remap = 1;
  dim3 grid(p.nq * 8 * nseq);
  float us = time_it([&] { attn_kernel<h16_t, 64, ABL><<<grid, 256>>>(p); });
  CK(hipDeviceSynchronize());
  printf("  abl=%2d %-34s %8.1f us  %7.1f TF\n", ABL, name, us, gf / us * 1e-3);
}
int main() {
  const int d = 512, T = 600;
  for (int nseq : {16, 64}) for (int S : {600, 2000}) {
    if (nseq == 64 && S == 600) continue;
    const int Sld = (S + 63) / 64 * 64;
    h16_t *q, *k, *vt, *o;
    CK(hipMalloc(&q, (size_t)nseq * T * d * 2)); CK(hipMalloc(&k, ((size_t)nseq * Sld + 64) * d * 2));
    CK(hipMalloc(&vt, (size_t)nseq * d * Sld * 2)); CK(hipMalloc(&o, (size_t)nseq * T * d * 2));
    std::vector<uint16_t> h((size_t)nseq * Sld * d + 64 * d);
    for (auto& v : h) v = 0x3c00 + (rand() & 0x7ff) - ((rand() & 1) << 15);
    CK(hipMemcpy(k, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(vt, h.data(), (size_t)nseq * d * Sld * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(q, h.data(), (size_t)nseq * T * d * 2, hipMemcpyHostToDevice));
    AttnP a; memset(&a, 0, sizeof(a));
    a.Q = q; a.q_seq_stride = (int64_t)T * d; a.ldq = d; a.K = k; a.k_slot_stride = (int64_t)Sld * d; a.ldk = d;
    a.VT = vt; a.vt_slot_stride = (int64_t)d * Sld; a.ldvt = Sld; a.O = o; a.o_seq_stride = (int64_t)T * d; a.ldo = d;
    a.tail_mod = 1; a.Tq = T; a.S_main = S; a.S_tail = 0; a.scale_log2e = 1.4426950408889634f / 8.0f;
    const double gf = 4.0 * nseq * 8 * T * (double)S * 64 * 1e-9;
    printf("nseq=%d T=%d S=%d (%.1f GF)\n", nseq, T, S, gf);
    run<0>("full", a, nseq, gf);
    run<1>("no exp", a, nseq, gf);
    run<2>("no max", a, nseq, gf);
    run<3>("no exp, no max", a, nseq, gf);
    run<4>("no staging/barriers", a, nseq, gf);
    run<8>("no PV mfma", a, nseq, gf);
    run<16>("no QK mfma", a, nseq, gf);
    run<24>("no mfma", a, nseq, gf);
    run<7>("no exp/max/staging", a, nseq, gf);
    run<32>("barrier, no DMA wait", a, nseq, gf);
    run<64>("DMA wait, no barrier", a, nseq, gf);
    run<31>("nothing but loads+glue", a, nseq, gf);
    CK(hipFree(q)); CK(hipFree(k)); CK(hipFree(vt)); CK(hipFree(o));
  }
  return 0;
}
