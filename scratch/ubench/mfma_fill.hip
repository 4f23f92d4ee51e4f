// How many single-issue VALU "fillers" hide under one MFMA in ONE wave's in-order stream, for the two MFMA shapes, at 1 and 2 waves
// per SIMD?  (round 6: the attention tile body is issue-bound; kernels_attn3.h)
//   hipcc --offload-arch=gfx950 -O3 scratch/ubench/mfma_fill.hip -o scratch/ubench/mfma_fill
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NF, int TRANS>   // NF fillers per MFMA; TRANS of them v_exp_f32
__device__ __forceinline__ void fillers(float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    if (i < TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i & 7]));
    else asm volatile("v_add_f32 %0, %0, %0" : "+v"(f[i & 7]));
  }
}
template <int SHAPE, int NF, int TRANS>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  h16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.01f); }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = threadIdx.x * 1e-3f + i;
  f32x4 c4[8];
  f32x16 c16[4];
  for (int i = 0; i < 8; ++i) c4[i] = f32x4{0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) c16[i][e] = 0;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (SHAPE == 16) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c4[i]) : "v"(a), "v"(b));
        fillers<NF, TRANS>(f);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c16[i]) : "v"(a), "v"(b));
        fillers<NF, TRANS>(f);
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += c4[i][0] + f[i];
  for (int i = 0; i < 4; ++i) s += c16[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int SHAPE, int NF, int TRANS>
void run(int threads) {
  float* out; long long* cyc;
  CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 8));
  const int iters = 2000;
  k<SHAPE, NF, TRANS><<<256, threads>>>(out, cyc, iters);
  k<SHAPE, NF, TRANS><<<256, threads>>>(out, cyc, iters);
  CK(hipDeviceSynchronize());
  long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const int nm = SHAPE == 16 ? 8 : 4;
  // s_memtime / readcyclecounter ticks at a constant 100 MHz on some parts: report raw ticks per MFMA AND wall via events would be better; here shader clock counter
  printf("shape %2d  %d waves/SIMD  fillers/MFMA %2d (%d exp): %7.2f ticks per MFMA  (%.2f per 16x16x32-equivalent)\n", SHAPE, threads / 256, NF, TRANS,
         (double)c / iters / nm, (double)c / iters / nm / (SHAPE == 16 ? 1 : 2));
  CK(hipFree(out)); CK(hipFree(cyc));
}
int main() {
  for (int th = 256; th <= 512; th += 256) {
    run<16, 0, 0>(th); run<16, 1, 0>(th); run<16, 2, 0>(th); run<16, 3, 0>(th); run<16, 3, 1>(th); run<16, 4, 1>(th); run<16, 6, 2>(th);
    run<32, 0, 0>(th); run<32, 2, 0>(th); run<32, 4, 0>(th); run<32, 5, 0>(th); run<32, 6, 2>(th); run<32, 8, 2>(th); run<32, 10, 3>(th); run<32, 12, 4>(th);
  }
  return 0;
}
