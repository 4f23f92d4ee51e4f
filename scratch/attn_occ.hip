// Occupancy / balance probe of the attention kernel (scratch; not product): the same kernel with grids that give every CU exactly
// 1, 2, 2.5, 3, 4, 6 workgroups -- is a CU's throughput saturated at 2 resident workgroups (then the 640-workgroup launch of the
// headline, 2.5 per CU, loses 17 % to imbalance) or does it still scale (then it is latency-bound and balance does not matter)?
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
template <int DH> void sweep(int nseq, int S) {
  const int d = 8 * DH, Tmax = 1536;
  const int Sld = (S + 63) / 64 * 64;
  h16_t *q, *k, *vt, *o;
  CK(hipMalloc(&q, (size_t)nseq * Tmax * d * 2)); CK(hipMalloc(&k, ((size_t)nseq * Sld + 64) * d * 2));
  CK(hipMalloc(&vt, (size_t)nseq * d * Sld * 2)); CK(hipMalloc(&o, (size_t)nseq * Tmax * d * 2));
  std::vector<uint16_t> h((size_t)nseq * (Sld > Tmax ? Sld : Tmax) * d + 64 * d);
  for (auto& v : h) v = 0x3800 + (rand() & 0x7ff) - ((rand() & 1) << 15);
  CK(hipMemcpy(k, h.data(), ((size_t)nseq * Sld + 64) * d * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(vt, h.data(), (size_t)nseq * d * Sld * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(q, h.data(), (size_t)nseq * Tmax * d * 2, hipMemcpyHostToDevice));
  printf("dh=%d nseq=%d S=%d: workgroups per CU -> us per launch, us per (workgroup per CU)\n", DH, nseq, S);
  const int pairs = nseq * 8;
  for (int T : {128 * 256 / pairs, 128 * 512 / pairs, 600, 128 * 768 / pairs, 128 * 1024 / pairs, 128 * 1536 / pairs, 128 * 2304 / pairs, 128 * 3072 / pairs}) {
    if (T < 128 || T > Tmax) continue;
    AttnP a; memset(&a, 0, sizeof(a));
    a.Q = q; a.q_seq_stride = (int64_t)T * d; a.ldq = d; a.K = k; a.k_slot_stride = (int64_t)Sld * d; a.ldk = d;
    a.VT = vt; a.vt_slot_stride = (int64_t)d * Sld; a.ldvt = Sld; a.O = o; a.o_seq_stride = (int64_t)T * d; a.ldo = d;
    a.tail_mod = 1; a.Tq = T; a.S_main = S; a.S_tail = 0; a.scale_log2e = 1.4426950408889634f / sqrtf((float)DH);
    a.nq = (T + 127) / 128; a.nheads = 8; a.nseq = nseq; a.xcd_remap = 1;
    dim3 grid(a.nq * 8 * nseq);
    float us = time_it([&] { attn_kernel<h16_t, DH><<<grid, 256>>>(a); });
    CK(hipDeviceSynchronize());
    const double wpc = grid.x / 256.0;
    const double gf = 4.0 * nseq * 8 * (double)T * S * DH * 1e-9;
    printf("  T=%5d grid=%5d  %.2f per CU  %8.1f us  %6.1f us/(wg/CU)  %7.1f TF\n", T, grid.x, wpc, us, us / wpc, gf / us * 1e3);
  }
  CK(hipFree(q)); CK(hipFree(k)); CK(hipFree(vt)); CK(hipFree(o));
}
#ifdef ATTN_QT_PATCH   // needs QT as a template parameter of attn_kernel (a three-line patch: template <..., int QTP = 2>, QT = QTP, __launch_bounds__(64 * NWV, QTP == 2 ? 3 : 2))
// 64 queries per wave (QTP = 4, two waves per workgroup, two per SIMD) against the shipped 32 (four waves per workgroup, three per SIMD)
template <int DH, int NWV, int QTP> void qt_shape(int nseq, int T, int S) {
  const int d = 8 * DH;
  const int Sld = (S + 63) / 64 * 64;
  h16_t *q, *k, *vt, *o;
  CK(hipMalloc(&q, (size_t)nseq * T * d * 2)); CK(hipMalloc(&k, ((size_t)nseq * Sld + 64) * d * 2));
  CK(hipMalloc(&vt, (size_t)nseq * d * Sld * 2)); CK(hipMalloc(&o, (size_t)nseq * T * d * 2));
  std::vector<uint16_t> h((size_t)nseq * (Sld > T ? Sld : T) * d + 64 * d);
  for (auto& v : h) v = 0x3800 + (rand() & 0x7ff) - ((rand() & 1) << 15);
  CK(hipMemcpy(k, h.data(), ((size_t)nseq * Sld + 64) * d * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(vt, h.data(), (size_t)nseq * d * Sld * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(q, h.data(), (size_t)nseq * T * d * 2, hipMemcpyHostToDevice));
  AttnP a; memset(&a, 0, sizeof(a));
  a.Q = q; a.q_seq_stride = (int64_t)T * d; a.ldq = d; a.K = k; a.k_slot_stride = (int64_t)Sld * d; a.ldk = d;
  a.VT = vt; a.vt_slot_stride = (int64_t)d * Sld; a.ldvt = Sld; a.O = o; a.o_seq_stride = (int64_t)T * d; a.ldo = d;
  a.tail_mod = 1; a.Tq = T; a.S_main = S; a.S_tail = 0; a.scale_log2e = 1.4426950408889634f / sqrtf((float)DH);
  constexpr int BQ = NWV * QTP * 16;
  a.nq = (T + BQ - 1) / BQ; a.nheads = 8; a.nseq = nseq; a.xcd_remap = 1;
  dim3 grid(a.nq * 8 * nseq);
  float us = time_it([&] { attn_kernel<h16_t, DH, 0, NWV, QTP><<<grid, 64 * NWV>>>(a); });
  CK(hipDeviceSynchronize());
  const double gf = 4.0 * nseq * 8 * (double)T * S * DH * 1e-9;
  printf("  dh=%d nseq=%3d T=%d S=%4d  %d waves x %2d queries  grid=%5d  %8.1f us  %7.1f TF\n", DH, nseq, T, S, NWV, QTP * 16, grid.x, us, gf / us * 1e3);
  CK(hipFree(q)); CK(hipFree(k)); CK(hipFree(vt)); CK(hipFree(o));
}
#endif
#ifdef ATTN_KS_PATCH   // needs scratch/attn_keysplit_experiment.patch applied to the kernel header
// key split (attn_kernel KS = 2) against the plain form: time and output difference at the shapes of the product
template <int KS> float run_ks(AttnP a, int nseq, h16_t* o) {
  a.O = o;
  dim3 grid(a.nq * 8 * nseq * KS);
  float us = time_it([&] { attn_kernel<h16_t, 64, 0, 4, KS><<<grid, 256>>>(a); });
  CK(hipDeviceSynchronize());
  return us;
}
void ks_shape(int nseq, int T, int S) {
  const int d = 512;
  const int Sld = (S + 63) / 64 * 64;
  h16_t *q, *k, *vt, *o1, *o2; float* part; int* sync;
  CK(hipMalloc(&q, (size_t)nseq * T * d * 2)); CK(hipMalloc(&k, ((size_t)nseq * Sld + 64) * d * 2));
  CK(hipMalloc(&vt, (size_t)nseq * d * Sld * 2)); CK(hipMalloc(&o1, (size_t)nseq * T * d * 2)); CK(hipMalloc(&o2, (size_t)nseq * T * d * 2));
  const int nq = (T + 127) / 128;
  const size_t waves = (size_t)nseq * 8 * nq * 4;
  CK(hipMalloc(&part, waves * ATTN_PART_FLOATS(64) * 4)); CK(hipMalloc(&sync, waves * 8)); CK(hipMemset(sync, 0, waves * 8));
  std::vector<uint16_t> h((size_t)nseq * (Sld > T ? Sld : T) * d + 64 * d);
  for (auto& v : h) v = 0x3400 + (rand() & 0xfff) - ((rand() & 1) << 15);
  CK(hipMemcpy(k, h.data(), ((size_t)nseq * Sld + 64) * d * 2, hipMemcpyHostToDevice));
  for (auto& v : h) v = 0x3400 + (rand() & 0xfff) - ((rand() & 1) << 15);
  CK(hipMemcpy(vt, h.data(), (size_t)nseq * d * Sld * 2, hipMemcpyHostToDevice));
  for (auto& v : h) v = 0x3800 + (rand() & 0xfff) - ((rand() & 1) << 15);
  CK(hipMemcpy(q, h.data(), (size_t)nseq * T * d * 2, hipMemcpyHostToDevice));
  AttnP a; memset(&a, 0, sizeof(a));
  a.Q = q; a.q_seq_stride = (int64_t)T * d; a.ldq = d; a.K = k; a.k_slot_stride = (int64_t)Sld * d; a.ldk = d;
  a.VT = vt; a.vt_slot_stride = (int64_t)d * Sld; a.ldvt = Sld; a.o_seq_stride = (int64_t)T * d; a.ldo = d;
  a.tail_mod = 1; a.Tq = T; a.S_main = S; a.S_tail = 0; a.scale_log2e = 1.4426950408889634f / 8.0f;
  a.nq = nq; a.nheads = 8; a.nseq = nseq; a.xcd_remap = 1; a.part = part; a.part_sync = sync;
  CK(hipMemset(o1, 0, (size_t)nseq * T * d * 2)); CK(hipMemset(o2, 0xff, (size_t)nseq * T * d * 2));
  const float t1 = run_ks<1>(a, nseq, o1), t2 = run_ks<2>(a, nseq, o2), t1b = run_ks<1>(a, nseq, o1), t2b = run_ks<2>(a, nseq, o2);
  std::vector<_Float16> h1((size_t)nseq * T * d), h2(h1.size());
  CK(hipMemcpy(h1.data(), o1, h1.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), o2, h2.size() * 2, hipMemcpyDeviceToHost));
  double num = 0, den = 0, mx = 0; size_t bad = 0;
  for (size_t i = 0; i < h1.size(); ++i) {
    const double x = (double)h1[i], y = (double)h2[i];
    if (!(y == y)) ++bad;
    num += (x - y) * (x - y); den += x * x; if (fabs(x - y) > mx) mx = fabs(x - y);
  }
  std::vector<int> hs(waves * 2); CK(hipMemcpy(hs.data(), sync, waves * 8, hipMemcpyDeviceToHost));
  int dirty = 0; for (int v : hs) dirty += v != 0;
  printf("  nseq=%3d T=%d S=%4d (%.2f workgroups per CU): plain %7.1f / %7.1f us   key split %7.1f / %7.1f us   rel l2 %.2e max %.2e nan %zu  sync words left set %d\n",
         nseq, T, S, nseq * 8 * nq / 256.0, t1, t1b, t2, t2b, sqrt(num / (den + 1e-30)), mx, bad, dirty);
  CK(hipFree(q)); CK(hipFree(k)); CK(hipFree(vt)); CK(hipFree(o1)); CK(hipFree(o2)); CK(hipFree(part)); CK(hipFree(sync));
}
#endif
int main(int argc, char** argv) {
#ifdef ATTN_KS_PATCH
  if (argc > 1) {
    for (int S : {2000, 1998, 600, 256}) for (int nseq : {16, 8, 4, 2, 20, 32}) ks_shape(nseq, 600, S);
    ks_shape(4, 240, 2000); ks_shape(16, 240, 240); ks_shape(16, 150, 2000);
    return 0;
  }
#endif
#ifdef ATTN_QT_PATCH
  if (argc > 1 && !strcmp(argv[1], "qt")) {
    for (int rep = 0; rep < 2; ++rep)
      for (int S : {2000, 600})
        for (int nseq : {16, 64}) {
          qt_shape<64, 4, 2>(nseq, 600, S); qt_shape<64, 2, 4>(nseq, 600, S);
        }
    for (int S : {2000, 600}) { qt_shape<32, 4, 2>(32, 600, S); qt_shape<32, 2, 4>(32, 600, S); }
    return 0;
  }
#endif
  sweep<64>(16, 2000);
  sweep<64>(16, 600);
  sweep<32>(32, 2000);
  sweep<32>(32, 600);
  return 0;
}
