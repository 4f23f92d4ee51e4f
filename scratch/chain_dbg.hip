// Localise the d=256 / MT=2 disagreement between the 4- and 8-wave chain shapes (scratch; not product).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/chain_dbg.hip -o scratch/chain_dbg
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <random>
#include "../audio2photoreal_amd/csrc/kernels_chain.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
constexpr int D = 256;
template <int MT, int MODE, int ABL, int NW>
std::vector<float> run(ChainP p, const float* x0, size_t xbytes, const h16_t* s4, const h16_t* s8) {
  CK(hipMemcpy(p.x, x0, xbytes, hipMemcpyDeviceToDevice));
  p.stream = NW == 8 ? s8 : s4;
  const int grid = (p.M + 16 * MT - 1) / (16 * MT);
  chain_kernel<D, MT, MODE, ABL, NW><<<grid, 64 * NW>>>(p);
  CK(hipDeviceSynchronize());
  std::vector<float> h(xbytes / 4);
  CK(hipMemcpy(h.data(), p.x, xbytes, hipMemcpyDeviceToHost));
  return h;
}
static void cmp(const char* name, const std::vector<float>& a, const std::vector<float>& b, int M) {
  int bad = 0, first = -1; double mx = 0; int rows = 0, lastrow = -1;
  for (size_t i = 0; i < a.size(); ++i)
    if (memcmp(&a[i], &b[i], 4)) { ++bad; if (first < 0) first = (int)i; double d = fabs((double)a[i] - b[i]); if (d > mx) mx = d; int r = (int)(i / D); if (r != lastrow) { ++rows; lastrow = r; } }
  printf("  %-44s %s  differing elements %d in %d rows, first row %d col %d, max |diff| %.3e\n", name, bad ? "DIFF" : "same", bad, rows,
         first < 0 ? -1 : first / D, first < 0 ? -1 : first % D, mx);
}
int main() {
  const int M = 2688, T = 448;
  std::mt19937 rng(1234); std::normal_distribution<float> nd(0.f, 1.f);
  const int nst = 8 + 2 * 4 * 4 + 64 + 200;   // generous: POST at d=256 consumes 2*4 + 8*(4+4) (+ pre work) stages
  std::vector<float> hx((size_t)M * D), hv(8192), hf((size_t)8 * 4 * D), haux(4096);
  for (auto& v : hx) v = nd(rng);
  for (auto& v : hv) v = 0.1f * nd(rng);
  for (int i = 512; i < 768; ++i) hv[i] = 1.f + 0.1f * nd(rng);     // lnA gamma
  for (auto& v : hf) v = 0.2f * nd(rng);
  for (auto& v : haux) v = 0.05f * nd(rng);
  std::vector<uint16_t> hain((size_t)M * D), hw((size_t)nst * 128 * 64);
  for (auto& v : hain) v = f2bf(nd(rng));
  for (auto& v : hw) v = f2bf(0.06f * nd(rng));
  float *x, *x0, *aux, *vec, *film; h16_t *ain, *w, *s4, *s8, *qk, *vt; float2* cs; ChainPackDesc* dd;
  CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&x0, hx.size() * 4)); CK(hipMalloc(&aux, haux.size() * 4)); CK(hipMalloc(&vec, hv.size() * 4));
  CK(hipMalloc(&film, hf.size() * 4)); CK(hipMalloc(&ain, hain.size() * 2)); CK(hipMalloc(&w, hw.size() * 2));
  CK(hipMalloc(&s4, (size_t)(nst + 8) * 16384)); CK(hipMalloc(&s8, (size_t)(nst + 8) * 16384));
  CK(hipMalloc(&qk, (size_t)M * 2 * D * 2)); CK(hipMalloc(&vt, (size_t)M * D * 2 + (1 << 20))); CK(hipMalloc(&cs, (size_t)640 * 128 * 8));
  CK(hipMemcpy(x0, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(aux, haux.data(), haux.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(vec, hv.data(), hv.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(film, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(ain, hain.data(), hain.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(cs, 0, (size_t)640 * 128 * 8)); CK(hipMemset(s4, 0, (size_t)(nst + 8) * 16384)); CK(hipMemset(s8, 0, (size_t)(nst + 8) * 16384));
  // both slice layouts of the SAME logical weights: stage i = rows [0,128) x k [0,64) of a [128, 64] block at w + i*8192
  std::vector<ChainPackDesc> descs(nst);
  for (int i = 0; i < nst; ++i) descs[i] = {w + (size_t)i * 8192, 64, 0, 0, 128};
  CK(hipMalloc(&dd, descs.size() * sizeof(ChainPackDesc))); CK(hipMemcpy(dd, descs.data(), descs.size() * sizeof(ChainPackDesc), hipMemcpyHostToDevice));
  chain_pack_kernel<<<nst, 256>>>(dd, s4, 4); chain_pack_kernel<<<nst, 256>>>(dd, s8, 8); CK(hipDeviceSynchronize());
  ChainP p; memset(&p, 0, sizeof(p));
  p.M = M; p.rows_per_seq = T; p.aux_kb = 4; p.x = x; p.aux = aux; p.ain = ain; p.ld_ain = D; p.has_next = 0;
  p.bias_o = vec; p.film_o = film; p.film_seq_stride = 4 * D; p.film_shift_off = D; p.lnA_g = vec + 512; p.lnA_b = vec + 1024;
  p.q_out = qk; p.ld_q = D; p.bias_2 = vec + 1536; p.film_f = film + 2 * D; p.lnB_g = vec + 2048; p.lnB_b = vec + 2560;
  p.qk_out = qk; p.ld_qk = 2 * D; p.vt_out = vt; p.vt_seq_stride = (int64_t)D * 448; p.ld_vt = 448; p.cst = reinterpret_cast<const f32x4*>(cs); p.cs_npos = 448;
  const size_t xb = hx.size() * 4;
  for (int rep = 0; rep < 3; ++rep) {
    printf("rep %d\n", rep);
    auto a = run<2, CHAIN_POST, 0, 4>(p, x0, xb, s4, s8);
    cmp("POST full          NW8/MT2 vs NW4/MT2", run<2, CHAIN_POST, 0, 8>(p, x0, xb, s4, s8), a, M);
    cmp("POST full          NW4/MT3 vs NW4/MT2", run<3, CHAIN_POST, 0, 4>(p, x0, xb, s4, s8), a, M);
    cmp("POST full          NW8/MT3 vs NW4/MT2", run<3, CHAIN_POST, 0, 8>(p, x0, xb, s4, s8), a, M);
    cmp("POST full          NW4/MT2 again     ", run<2, CHAIN_POST, 0, 4>(p, x0, xb, s4, s8), a, M);
    auto b = run<2, CHAIN_POST, 256, 4>(p, x0, xb, s4, s8);
    cmp("stop after out_proj NW8/MT2 vs NW4/MT2", run<2, CHAIN_POST, 256, 8>(p, x0, xb, s4, s8), b, M);
    auto c = run<2, CHAIN_POST, 512, 4>(p, x0, xb, s4, s8);
    cmp("out_proj discarded  NW8/MT2 vs NW4/MT2", run<2, CHAIN_POST, 512, 8>(p, x0, xb, s4, s8), c, M);
    auto d = run<2, CHAIN_POST, 1024, 4>(p, x0, xb, s4, s8);
    cmp("FFN discarded       NW8/MT2 vs NW4/MT2", run<2, CHAIN_POST, 1024, 8>(p, x0, xb, s4, s8), d, M);
    auto e = run<2, CHAIN_MID, 0, 4>(p, x0, xb, s4, s8);
    cmp("MID                 NW8/MT2 vs NW4/MT2", run<2, CHAIN_MID, 0, 8>(p, x0, xb, s4, s8), e, M);
  }
  return 0;
}
