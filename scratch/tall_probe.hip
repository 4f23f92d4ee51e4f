// Probe for the round-3 chain-kernel restructure (scratch; not product): cycles per 16 KiB weight stage of a [16*MT rows] x
// [128 cols] x [64 k] GEMM step for panel heights MT = 3..6, 4 or 8 waves, and two ways of bringing the weights in:
//   WPATH 0: LDS-DMA into a wave-private ring (what kernels_chain.h does today), fragments read back with ds_read_b128
//   WPATH 1: global_load_dwordx4 straight into VGPRs, PF stages ahead (no LDS write, no W fragment reads, no ring in LDS)
// The A panel sits in LDS in the product's swizzled layout; its fragment reads are software-pipelined one stage ahead and, with
// FRONT = 1, issued in the first half of the stage's MFMAs so their latency passes under the second half.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/tall_probe.hip -o scratch/tall_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef _Float16 h16;
typedef __attribute__((ext_vector_type(8))) h16 h16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int STAGES = 256;

template <int MT, int NW, int WPATH, int PF, int FRONT>
__global__ __launch_bounds__(64 * NW, 1) void probe(const char* __restrict__ src, unsigned long long* out, float* sink) {
  constexpr int CW = 128 / NW, NJ = CW / 16, PCS = CW / 8, BM = 16 * MT, D = 512;
  constexpr int NS = 4;                                   // ring slots (WPATH 0)
  constexpr int PANEL_B = BM * D * 2;
  __shared__ __attribute__((aligned(16))) char lds[PANEL_B + (WPATH == 0 ? NS * 16384 : 16)];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
  for (int i = threadIdx.x * 16; i < PANEL_B; i += 64 * NW * 16) *reinterpret_cast<float4*>(lds + i) = make_float4(1e-3f, 2e-3f, 3e-3f, 4e-3f);
  char* ring = lds + PANEL_B + wid * (NS * PCS * 1024);
  const char* wsrc = src + (size_t)wid * (PCS * 1024) + lane * 16;   // this wave's slice of every 16 KiB stage
  f32x4 acc[MT][NJ];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[mt][j] = f32x4{0, 0, 0, 0};
  __syncthreads();
  auto load_a = [&](h16x8(&a)[2][MT], int s) __attribute__((always_inline)) {
    const int ks = s & 7;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        a[kk][mt] = *reinterpret_cast<const h16x8*>(lds + ((mt * 16 + l15) * D + (((ks * 8 + kk * 4 + g) ^ l15) << 3)) * 2);
  };
  auto mma = [&](const h16x8(&a)[2][MT], const h16x8(&w)[2][NJ]) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[kk][j], a[kk][mt], acc[mt][j], 0, 0, 0);
  };
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long r0 = wall_clock64();
  if constexpr (WPATH == 1) {
    // direct: w[PF] stages of fragments in flight; the compiler counts the vmcnt waits (every load has its own register)
    h16x8 w[PF][2][NJ];
    h16x8 a[2][2][MT];
    auto load_w = [&](h16x8(&d)[2][NJ], int s) __attribute__((always_inline)) {
      const char* p = wsrc + (size_t)(s & 127) * 16384;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < NJ; ++j) d[kk][j] = *reinterpret_cast<const h16x8*>(p + (kk * NJ + j) * 1024);
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) load_w(w[i], i);
    load_a(a[0], 0);
    for (int s0 = 0; s0 < STAGES; s0 += 2 * PF) {
#pragma unroll
      for (int u = 0; u < 2 * PF; ++u) {
        const int s = s0 + u;
        load_a(a[(u + 1) & 1], s + 1);
        mma(a[u & 1], w[u % PF]);
        load_w(w[u % PF], s + PF);
        if constexpr (FRONT) {
          // A reads first (2 per MFMA), the global loads behind them, the remaining MFMAs cover the latencies
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          }
#pragma unroll
          for (int i = 0; i < 2 * NJ; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < PF; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(w[i][kk][j]));
  } else {
    h16x8 a[2][2][MT], w[2][2][NJ];
    int issue = 0, consume = 0;
    const char* gsrc = wsrc;
    int sidx = 0;
    auto issue_stage = [&]() __attribute__((always_inline)) {
      const auto gp = (const __attribute__((address_space(1))) void*)(gsrc + (size_t)(sidx & 127) * 16384);
      const auto lp = (__attribute__((address_space(3))) void*)(ring + issue);
      __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
      __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 0);
      if constexpr (PCS == 4) {
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds(gp, lp, 16, 3072, 0);
      }
      ++sidx;
      issue = issue + PCS * 1024 == NS * PCS * 1024 ? 0 : issue + PCS * 1024;
    };
    auto stage_begin = [&]() __attribute__((always_inline)) -> const char* {
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PCS * (NS - 2)) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      issue_stage();
      const char* wb = ring + consume;
      consume = consume + PCS * 1024 == NS * PCS * 1024 ? 0 : consume + PCS * 1024;
      return wb;
    };
    auto load_w = [&](h16x8(&d)[2][NJ], const char* wb) __attribute__((always_inline)) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int wrow = j * 16 + l15;
          d[kk][j] = *reinterpret_cast<const h16x8*>(wb + (wrow * 64 + (((kk * 4 + g) ^ ((wrow >> 1) & 7)) << 3)) * 2);
        }
    };
#pragma unroll
    for (int i = 0; i < NS - 1; ++i) issue_stage();
    load_a(a[0], 0);
    load_w(w[0], stage_begin());
    for (int s0 = 0; s0 < STAGES; s0 += 2) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int s = s0 + u;
        const char* wb = stage_begin();
        load_a(a[(u + 1) & 1], s + 1);
        load_w(w[(u + 1) & 1], wb);
        mma(a[u & 1], w[u & 1]);
#pragma unroll
        for (int i = 0; i < PCS; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        if constexpr (FRONT) {
#pragma unroll
          for (int i = 0; i < MT + NJ; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 2 * MT + 2 * NJ; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = wall_clock64();
  float v = 0;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NJ; ++j) v += acc[mt][j][0] + acc[mt][j][3];
  if (v == 1.2345f) sink[0] = v;
  if (lane == 0) {
    out[(blockIdx.x * NW + wid) * 2 + 0] = t1 - t0;
    out[(blockIdx.x * NW + wid) * 2 + 1] = r1 - r0;
  }
}

template <int MT, int NW, int WPATH, int PF, int FRONT>
void run(const char* src, unsigned long long* dout, float* sink, int blocks) {
  std::vector<unsigned long long> h((size_t)blocks * NW * 2);
  for (int it = 0; it < 3; ++it) probe<MT, NW, WPATH, PF, FRONT><<<blocks, 64 * NW>>>(src, dout, sink);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0, rt = 0;
  for (size_t i = 0; i < h.size(); i += 2) { cyc += (double)h[i]; rt += (double)h[i + 1]; }
  cyc /= (h.size() / 2); rt /= (h.size() / 2);
  const double per = cyc / STAGES, mfma = 64.0 * MT;   // MFMA-bound floor: 16*MT rows x 128 cols x 64 k at 4069 FLOP/clk/CU
  printf("  blocks=%3d rows=%3d NW=%d %-10s PF=%d front=%d : %7.1f cyc/stage  %6.2f cyc/row  MFMA floor %5.0f (%4.1f %%)  %6.3f us/stage  clk %.2f GHz\n", blocks, 16 * MT, NW,
         WPATH ? "direct-W" : "LDS-DMA", PF, FRONT, per, per / (16 * MT), mfma, 100.0 * mfma / per, rt * 0.01 / STAGES, cyc / (rt * 10.0));
}

int main(int argc, char** argv) {
  char* src; unsigned long long* dout; float* sink;
  CK(hipMalloc(&src, (size_t)(128 + 16) * 16384));
  CK(hipMemset(src, 0x11, (size_t)(128 + 16) * 16384));
  CK(hipMalloc(&dout, 256 * 8 * 2 * 8));
  CK(hipMalloc(&sink, 64));
  for (int blocks : {1, 256}) {
    printf("blocks=%d\n", blocks);
    run<3, 4, 0, 1, 0>(src, dout, sink, blocks);   // today's loop
    run<3, 4, 0, 1, 1>(src, dout, sink, blocks);
    run<3, 8, 0, 1, 0>(src, dout, sink, blocks);
    run<3, 8, 0, 1, 1>(src, dout, sink, blocks);
    run<3, 4, 1, 3, 1>(src, dout, sink, blocks);
    run<3, 4, 1, 4, 1>(src, dout, sink, blocks);
    run<3, 4, 1, 4, 0>(src, dout, sink, blocks);
    run<3, 8, 1, 4, 1>(src, dout, sink, blocks);
    run<3, 8, 1, 4, 0>(src, dout, sink, blocks);
    run<4, 4, 0, 1, 1>(src, dout, sink, blocks);
    run<4, 4, 1, 4, 1>(src, dout, sink, blocks);
    run<4, 8, 1, 4, 1>(src, dout, sink, blocks);
    run<5, 4, 0, 1, 1>(src, dout, sink, blocks);
    run<5, 4, 1, 3, 1>(src, dout, sink, blocks);
    run<5, 4, 1, 4, 1>(src, dout, sink, blocks);
    run<5, 4, 1, 4, 0>(src, dout, sink, blocks);
    run<5, 8, 1, 4, 1>(src, dout, sink, blocks);
    run<6, 4, 0, 1, 1>(src, dout, sink, blocks);
    run<6, 4, 1, 3, 1>(src, dout, sink, blocks);
    run<6, 4, 1, 4, 1>(src, dout, sink, blocks);
    run<6, 4, 1, 4, 0>(src, dout, sink, blocks);
    run<6, 8, 1, 4, 1>(src, dout, sink, blocks);
  }
  return 0;
}
