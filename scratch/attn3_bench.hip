// attn3_kernel (kernels_attn3.h) against attn_kernel on the product's attention shapes: time per launch and output difference.
// attn3 uses a lazy softmax reference and pre-scaled Q: the outputs agree to operand rounding, not bit for bit.  Scratch, not product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DA2P_HALF scratch/attn3_bench.hip -o scratch/attn3_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../audio2photoreal_amd/csrc/kernels_attn3.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
template <typename F>
float time_it(F f, int iters = 20) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(e0)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters * 1e3f;
}
static uint16_t rnd_h(float scale) {
  _Float16 v = (_Float16)(scale * (2.0f * (rand() / (float)RAND_MAX) - 1.0f));
  uint16_t u; memcpy(&u, &v, 2); return u;
}
template <int DH> void shape(int nseq, int T, int S_main, int S_tail, int shared_slot0, int nt, float qk_amp = 1.5f) {
  const int H = 8, d = H * DH, S = S_main + S_tail, Sld = (S + 63) / 64 * 64;
  const int nslot = nseq + 1;
  h16_t *q, *k, *vt, *o1, *o2; float *kt, *vtl; int* stat;
  CK(hipMalloc(&q, (size_t)nseq * T * d * 2)); CK(hipMalloc(&k, (size_t)nslot * Sld * d * 2)); CK(hipMalloc(&vt, (size_t)nslot * d * Sld * 2));
  CK(hipMalloc(&o1, (size_t)nseq * T * d * 2)); CK(hipMalloc(&o2, (size_t)nseq * T * d * 2));
  CK(hipMalloc(&kt, (size_t)nseq * 2 * d * 4)); CK(hipMalloc(&vtl, (size_t)nseq * 2 * d * 4)); CK(hipMalloc(&stat, 8));
  std::vector<uint16_t> h((size_t)nslot * Sld * d);
  for (auto& v : h) v = rnd_h(qk_amp);
  CK(hipMemcpy(k, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  for (auto& v : h) v = rnd_h(1.0f);
  CK(hipMemcpy(vt, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  std::vector<uint16_t> hq((size_t)nseq * T * d);
  for (auto& v : hq) v = rnd_h(qk_amp);
  CK(hipMemcpy(q, hq.data(), hq.size() * 2, hipMemcpyHostToDevice));
  std::vector<float> ht((size_t)nseq * 2 * d);
  for (auto& v : ht) v = 2.0f * (rand() / (float)RAND_MAX) - 1.0f;
  CK(hipMemcpy(kt, ht.data(), ht.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(vtl, ht.data(), ht.size() * 4, hipMemcpyHostToDevice));
  AttnP a; memset(&a, 0, sizeof(a));
  a.Q = q; a.q_seq_stride = (int64_t)T * d; a.ldq = d; a.K = k; a.k_slot_stride = (int64_t)Sld * d; a.ldk = d;
  a.VT = vt; a.vt_slot_stride = (int64_t)d * Sld; a.ldvt = Sld; a.o_seq_stride = (int64_t)T * d; a.ldo = d;
  a.ktail = S_tail ? kt : nullptr; a.vtail = S_tail ? vtl : nullptr; a.tail_sample_stride = 2 * d; a.tail_row_stride = d;
  a.tail_mod = nseq; a.Tq = T; a.S_main = S_main; a.S_tail = S_tail; a.scale_log2e = 1.4426950408889634f / sqrtf((float)DH);
  a.slot_rule = shared_slot0 ? 3 : 1; a.slot_b = nseq / 2; a.kv_stream = nt;
  a.nheads = H; a.nseq = nseq; a.xcd_remap = 1; a.stat_max = stat;
  int sinit[2] = {(int)0x80000000, (int)0x80000000};
  AttnP a1 = a, a2 = a;
  a1.O = o1; a1.nq = (T + 127) / 128;
  a2.O = o2; a2.nq = (T + 319) / 320;
  CK(hipMemset(o1, 0, (size_t)nseq * T * d * 2)); CK(hipMemset(o2, 0xff, (size_t)nseq * T * d * 2));
  float t1 = 0, t2 = 0;
  for (int rep = 0; rep < 2; ++rep) {
    t1 = time_it([&] { attn_kernel<h16_t, DH><<<dim3(a1.nq * H * nseq), 256>>>(a1); });
    t2 = time_it([&] { attn3_kernel<DH><<<dim3(a2.nq * H * nseq), 256>>>(a2); });
  }
  int st1 = 0, st2 = 0;
  CK(hipMemcpy(stat, sinit, 4, hipMemcpyHostToDevice)); attn_kernel<h16_t, DH><<<dim3(a1.nq * H * nseq), 256>>>(a1); CK(hipMemcpy(&st1, stat, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(stat, sinit, 4, hipMemcpyHostToDevice)); attn3_kernel<DH><<<dim3(a2.nq * H * nseq), 256>>>(a2); CK(hipMemcpy(&st2, stat, 4, hipMemcpyDeviceToHost));
  CK(hipDeviceSynchronize());
  std::vector<_Float16> h1((size_t)nseq * T * d), h2(h1.size());
  CK(hipMemcpy(h1.data(), o1, h1.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), o2, h2.size() * 2, hipMemcpyDeviceToHost));
  double mx = 0, num = 0, den = 0; size_t nan = 0;
  for (size_t i = 0; i < h1.size(); ++i) {
    const double x = (double)h1[i], y = (double)h2[i];
    if (!(y == y)) { ++nan; continue; }
    num += (x - y) * (x - y); den += x * x;
    if (fabs(x - y) > mx) mx = fabs(x - y);
  }
  float f1, f2; memcpy(&f1, &st1, 4); memcpy(&f2, &st2, 4);   // (positive floats map to themselves under attn_ordered_int)
  const double gf = 4.0 * nseq * H * (double)T * S * DH * 1e-9;
  printf("dh=%d nseq=%3d T=%d S=%d+%d slot0=%d nt=%d amp=%.1f | attn %7.1f us %6.1f TF (%5d wg) | attn3 %7.1f us %6.1f TF (%4d wg) | x%.2f | rel-l2 %.3e max-abs %.3e nan %zu | logit max %.3f vs %.3f\n",
         DH, nseq, T, S_main, S_tail, shared_slot0, nt, qk_amp, t1, gf / t1 * 1e3, a1.nq * H * nseq, t2, gf / t2 * 1e3, a2.nq * H * nseq, t1 / t2,
         sqrt(num / (den + 1e-30)), mx, nan, f1, f2);
  CK(hipFree(q)); CK(hipFree(k)); CK(hipFree(vt)); CK(hipFree(o1)); CK(hipFree(o2)); CK(hipFree(kt)); CK(hipFree(vtl)); CK(hipFree(stat));
}
#ifdef ATTN3_ABL
template <int ABL> float abl_time(AttnP a, int H, int nseq) {
  return time_it([&] { attn3_kernel<64, ABL><<<dim3(a.nq * H * nseq), 256>>>(a); });
}
void ablate() {
  const int DH = 64, nseq = 16, T = 600, S_main = 2000, H = 8, d = H * DH, Sld = 2048;
  h16_t *q, *k, *vt, *o;
  CK(hipMalloc(&q, (size_t)nseq * T * d * 2)); CK(hipMalloc(&k, (size_t)(nseq + 1) * Sld * d * 2)); CK(hipMalloc(&vt, (size_t)(nseq + 1) * d * Sld * 2));
  CK(hipMalloc(&o, (size_t)nseq * T * d * 2));
  std::vector<uint16_t> h((size_t)(nseq + 1) * Sld * d);
  for (auto& v : h) v = rnd_h(1.0f);
  CK(hipMemcpy(k, h.data(), h.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(vt, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(q, h.data(), (size_t)nseq * T * d * 2, hipMemcpyHostToDevice));
  AttnP a; memset(&a, 0, sizeof(a));
  a.Q = q; a.q_seq_stride = (int64_t)T * d; a.ldq = d; a.K = k; a.k_slot_stride = (int64_t)Sld * d; a.ldk = d;
  a.VT = vt; a.vt_slot_stride = (int64_t)d * Sld; a.ldvt = Sld; a.O = o; a.o_seq_stride = (int64_t)T * d; a.ldo = d;
  a.tail_mod = nseq; a.Tq = T; a.S_main = S_main; a.S_tail = 0; a.scale_log2e = 1.4426950408889634f / 8.0f;
  a.slot_rule = 1; a.nheads = H; a.nseq = nseq; a.xcd_remap = 1; a.nq = (T + 319) / 320;
  printf("attn3 ablation (us): full %.1f | no exp %.1f | no max %.1f | no sum/cvt.. %.1f | no exp,max,sum %.1f | no PV %.1f | no QK %.1f | no MFMA %.1f | skeleton (no MFMA, no softmax) %.1f | skeleton no DMA %.1f | skeleton no DMA no barrier %.1f | full no DMA %.1f\n",
         abl_time<0>(a, H, nseq), abl_time<1>(a, H, nseq), abl_time<2>(a, H, nseq), abl_time<32>(a, H, nseq), abl_time<35>(a, H, nseq), abl_time<8>(a, H, nseq),
         abl_time<16>(a, H, nseq), abl_time<24>(a, H, nseq), abl_time<59>(a, H, nseq), abl_time<59 | 64>(a, H, nseq), abl_time<59 | 64 | 128>(a, H, nseq), abl_time<64>(a, H, nseq));
}
#endif
#ifndef A3_ABL
#define A3_ABL 0
#endif
#ifdef A3_WGSTAMPS
// per-workgroup timeline of one launch (wave 0 of every workgroup): when it started, how long its prologue / loop / epilogue took
template <int DH> void wg_timeline(const char* name, int nseq, int T, int S_main, int S_tail, int shared_slot0, int nt) {
  const int H = 8, d = H * DH, S = S_main + S_tail, Sld = (S + 63) / 64 * 64, nslot = nseq + 1;
  h16_t *q, *k, *vt, *o; float *kt, *vtl; long long* dbg;
  CK(hipMalloc(&q, (size_t)nseq * T * d * 2)); CK(hipMalloc(&k, (size_t)nslot * Sld * d * 2)); CK(hipMalloc(&vt, (size_t)nslot * d * Sld * 2));
  CK(hipMalloc(&o, (size_t)nseq * T * d * 2)); CK(hipMalloc(&kt, (size_t)nseq * 2 * d * 4)); CK(hipMalloc(&vtl, (size_t)nseq * 2 * d * 4));
  std::vector<uint16_t> h((size_t)nslot * Sld * d);
  for (auto& v : h) v = rnd_h(1.5f);
  CK(hipMemcpy(k, h.data(), h.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(vt, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(q, h.data(), (size_t)nseq * T * d * 2, hipMemcpyHostToDevice));
  CK(hipMemset(kt, 0, (size_t)nseq * 2 * d * 4)); CK(hipMemset(vtl, 0, (size_t)nseq * 2 * d * 4));
  AttnP a; memset(&a, 0, sizeof(a));
  a.Q = q; a.q_seq_stride = (int64_t)T * d; a.ldq = d; a.K = k; a.k_slot_stride = (int64_t)Sld * d; a.ldk = d;
  a.VT = vt; a.vt_slot_stride = (int64_t)d * Sld; a.ldvt = Sld; a.O = o; a.o_seq_stride = (int64_t)T * d; a.ldo = d;
  a.ktail = S_tail ? kt : nullptr; a.vtail = S_tail ? vtl : nullptr; a.tail_sample_stride = 2 * d; a.tail_row_stride = d;
  a.tail_mod = nseq; a.Tq = T; a.S_main = S_main; a.S_tail = S_tail; a.scale_log2e = 1.4426950408889634f / sqrtf((float)DH);
  a.slot_rule = shared_slot0 ? 3 : 1; a.slot_b = nseq / 2; a.kv_stream = nt;
  a.nheads = H; a.nseq = nseq; a.xcd_remap = 1; a.nq = (T + 319) / 320;
  const int nwg = a.nq * H * nseq;
  CK(hipMalloc(&dbg, (size_t)nwg * 64)); CK(hipMemset(dbg, 0, (size_t)nwg * 64));
  a.kv_slot = (const int*)dbg;
  const float us = time_it([&] { attn3_kernel<DH, A3_ABL><<<dim3(nwg), 256>>>(a); });
  std::vector<long long> hd((size_t)nwg * 8);
  CK(hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost));
  long long t0 = hd[0], t3 = hd[3];
  for (int w = 0; w < nwg; ++w) { if (hd[w * 8] < t0) t0 = hd[w * 8]; if (hd[w * 8 + 3] > t3) t3 = hd[w * 8 + 3]; }
  double s[5] = {0, 0, 0, 0, 0}, mx[5] = {0, 0, 0, 0, 0}, mn[5] = {1e30, 1e30, 1e30, 1e30, 1e30}, cyc = 0;
  for (int w = 0; w < nwg; ++w) {
    const long long* r = &hd[w * 8];
    const double v[5] = {(r[0] - t0) * 0.01, (r[1] - r[0]) * 0.01, (r[2] - r[1]) * 0.01, (r[3] - r[2]) * 0.01, (r[3] - t0) * 0.01};
    for (int i = 0; i < 5; ++i) { s[i] += v[i]; if (v[i] > mx[i]) mx[i] = v[i]; if (v[i] < mn[i]) mn[i] = v[i]; }
    cyc += (double)(r[5] - r[4]);
  }
  const int nsteps = (S + 63) / 64;
  printf("%s: %d wg, launch-to-launch %.1f us, first start -> last end %.1f us | per workgroup avg [min..max] us: start +%.2f [%.2f..%.2f]  prologue %.2f [%.2f..%.2f]  loop %.2f [%.2f..%.2f]  epilogue %.2f [%.2f..%.2f]  end +%.2f [%.2f..%.2f] | loop %.0f cycles = %.0f per tile step (%d), clock %.3f GHz\n",
         name, nwg, us, (t3 - t0) * 0.01, s[0] / nwg, mn[0], mx[0], s[1] / nwg, mn[1], mx[1], s[2] / nwg, mn[2], mx[2], s[3] / nwg, mn[3], mx[3], s[4] / nwg, mn[4], mx[4],
         cyc / nwg, cyc / nwg / nsteps, nsteps, cyc / nwg / (s[2] / nwg) * 1e-3);
  // by slot class (shared slot 0 = L2-resident; own slot = streamed)
  if (shared_slot0) {
    double sl[2] = {0, 0}; int n[2] = {0, 0};
    for (int w = 0; w < nwg; ++w) {
      const int xcd = w & 7, j = w >> 3, pair = (j / a.nq) * 8 + xcd, seq = pair / H;
      const int c = seq < a.slot_b ? 1 : 0;
      sl[c] += (hd[w * 8 + 2] - hd[w * 8 + 1]) * 0.01; ++n[c];
    }
    printf("    loop us by K/V source: shared slot 0 (L2) %.2f (%d wg) | own slot (streamed) %.2f (%d wg)\n", sl[0] / (n[0] ? n[0] : 1), n[0], sl[1] / (n[1] ? n[1] : 1), n[1]);
  }
  CK(hipFree(q)); CK(hipFree(k)); CK(hipFree(vt)); CK(hipFree(o)); CK(hipFree(kt)); CK(hipFree(vtl)); CK(hipFree(dbg));
}
#endif
int main(int argc, char** argv) {
#ifdef A3_WGSTAMPS
  if (argc > 1 && !strcmp(argv[1], "wg")) {
    wg_timeline<64>("B=8 cross", 16, 600, 1998, 2, 1, 1);
    wg_timeline<64>("B=8 self ", 16, 600, 600, 0, 0, 0);
    wg_timeline<64>("B=32 cross", 64, 600, 1998, 2, 1, 1);
    wg_timeline<64>("B=32 self ", 64, 600, 600, 0, 0, 0);
    return 0;
  }
#endif
#ifdef ATTN3_ABL
  if (argc > 1 && !strcmp(argv[1], "abl")) { ablate(); return 0; }
#endif
  if (argc > 1 && !strcmp(argv[1], "sk")) {   // skeleton experiments: tile DMA / barriers / cache policy (timing only)
    const int DH = 64, nseq = 16, T = 600, S_main = 2000, H = 8, d = H * DH, Sld = 2048;
    h16_t *q, *k, *vt, *o;
    CK(hipMalloc(&q, (size_t)nseq * T * d * 2)); CK(hipMalloc(&k, (size_t)(nseq + 1) * Sld * d * 2)); CK(hipMalloc(&vt, (size_t)(nseq + 1) * d * Sld * 2));
    CK(hipMalloc(&o, (size_t)nseq * T * d * 2));
    std::vector<uint16_t> h((size_t)(nseq + 1) * Sld * d);
    for (auto& v : h) v = rnd_h(1.0f);
    CK(hipMemcpy(k, h.data(), h.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(vt, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(q, h.data(), (size_t)nseq * T * d * 2, hipMemcpyHostToDevice));
    AttnP a; memset(&a, 0, sizeof(a));
    a.Q = q; a.q_seq_stride = (int64_t)T * d; a.ldq = d; a.K = k; a.k_slot_stride = (int64_t)Sld * d; a.ldk = d;
    a.VT = vt; a.vt_slot_stride = (int64_t)d * Sld; a.ldvt = Sld; a.O = o; a.o_seq_stride = (int64_t)T * d; a.ldo = d;
    a.tail_mod = nseq; a.Tq = T; a.S_main = S_main; a.S_tail = 0; a.scale_log2e = 1.4426950408889634f / 8.0f;
    a.nheads = H; a.nseq = nseq; a.xcd_remap = 1; a.nq = (T + 319) / 320;
    for (int rule = 1; rule <= 3; rule += 2)
      for (int nt = 0; nt < 2; ++nt) {
        a.slot_rule = rule; a.slot_b = nseq / 2; a.kv_stream = nt;
        const float t0 = time_it([&] { attn3_kernel<64, 0><<<dim3(a.nq * H * nseq), 256>>>(a); });
        const float t1 = time_it([&] { attn3_kernel<64, 64><<<dim3(a.nq * H * nseq), 256>>>(a); });
        const float t2 = time_it([&] { attn3_kernel<64, 128><<<dim3(a.nq * H * nseq), 256>>>(a); });
        const float t3 = time_it([&] { attn3_kernel<64, 192><<<dim3(a.nq * H * nseq), 256>>>(a); });
        printf("A3X=%d slot_rule=%d nt=%d: full skeleton %.1f us | no tile DMA %.1f | no barrier %.1f | neither %.1f\n", A3X, rule, nt, t0, t1, t2, t3);
      }
    return 0;
  }
#ifdef A3_STAMPS
  if (argc > 1 && !strcmp(argv[1], "st")) {
    const int DH = 64, nseq = 16, T = 600, S_main = 1984, H = 8, d = H * DH, Sld = 2048;   // 31 full tiles, no partial tile
    h16_t *q, *k, *vt, *o; long long* dbg;
    CK(hipMalloc(&q, (size_t)nseq * T * d * 2)); CK(hipMalloc(&k, (size_t)(nseq + 1) * Sld * d * 2)); CK(hipMalloc(&vt, (size_t)(nseq + 1) * d * Sld * 2));
    CK(hipMalloc(&o, (size_t)nseq * T * d * 2)); CK(hipMalloc(&dbg, 64)); CK(hipMemset(dbg, 0, 64));
    std::vector<uint16_t> h((size_t)(nseq + 1) * Sld * d);
    for (auto& v : h) v = rnd_h(1.5f);
    CK(hipMemcpy(k, h.data(), h.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(vt, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(q, h.data(), (size_t)nseq * T * d * 2, hipMemcpyHostToDevice));
    AttnP a; memset(&a, 0, sizeof(a));
    a.Q = q; a.q_seq_stride = (int64_t)T * d; a.ldq = d; a.K = k; a.k_slot_stride = (int64_t)Sld * d; a.ldk = d;
    a.VT = vt; a.vt_slot_stride = (int64_t)d * Sld; a.ldvt = Sld; a.O = o; a.o_seq_stride = (int64_t)T * d; a.ldo = d;
    a.tail_mod = nseq; a.Tq = T; a.S_main = S_main; a.S_tail = 0; a.scale_log2e = 1.4426950408889634f / 8.0f;
    a.slot_rule = 3; a.slot_b = nseq / 2; a.kv_stream = 1; a.kv_slot = (const int*)dbg;
    a.nheads = H; a.nseq = nseq; a.xcd_remap = 1; a.nq = (T + 319) / 320;
    const float t0 = time_it([&] { attn3_kernel<64><<<dim3(a.nq * H * nseq), 256>>>(a); });
    long long hd[8]; CK(hipMemcpy(hd, dbg, 64, hipMemcpyDeviceToHost));
    const char* nm[8] = {"softmax of query tile 0", "five MFMA groups", "landed nops + tail maxima + DMA wait", "barrier", "tile DMA issue", "PROLOGUE (once, not per step: x31)", "loop control + must_move + fragment wait", "EPILOGUE (once: x31)"};
    printf("stamped kernel %.1f us; per step (31 steps), workgroup 0 wave 0, cycles:\n", t0);
    long long tot = 0;
    for (int i = 0; i < 8; ++i) tot += hd[i];
    for (int i = 0; i < 8; ++i) if (hd[i]) printf("  %-46s %8.1f  (%4.1f %%)\n", nm[i], hd[i] / 31.0, 100.0 * hd[i] / tot);
    printf("  total %.1f cycles per step\n", tot / 31.0);
    return 0;
  }
#endif
  if (argc > 1 && !strcmp(argv[1], "small")) {   // fewer sequences than the headline: where does the 4 x 32-query kernel win back?
    for (int nseq : {2, 4, 6, 8, 12, 16}) { shape<64>(nseq, 600, 1998, 2, 1, 1); shape<64>(nseq, 600, 600, 0, 0, 0); }
    return 0;
  }
  if (argc > 1) { shape<64>(16, 600, 1998, 2, 1, 1); return 0; }
  shape<64>(16, 600, 1998, 2, 1, 1);
  shape<64>(16, 600, 600, 0, 0, 0);
  shape<64>(64, 600, 1998, 2, 1, 1);
  shape<64>(64, 600, 600, 0, 0, 0);
  shape<32>(32, 600, 1998, 2, 1, 1);
  shape<32>(32, 600, 600, 0, 0, 0);
  shape<64>(16, 600, 1998, 2, 1, 1, 4.0f);     // peaked rows: the lazy reference has to move (logits ~ +-20)
  shape<64>(2, 33, 20, 0, 0, 0);
  shape<64>(2, 100, 77, 2, 0, 0);
  shape<64>(8, 240, 127, 2, 1, 0);
  shape<64>(8, 240, 798, 2, 1, 1);
  shape<32>(8, 321, 192, 0, 0, 0);
  shape<32>(8, 640, 254, 2, 1, 0);
  shape<64>(4, 150, 150, 0, 0, 0);
  return 0;
}
