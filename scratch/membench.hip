// Box characterisation (scratch): HBM / MALL / L2 read bandwidth + a fixed-work MFMA loop, to correlate with the
// box-to-box spread of the memory-bound chain kernels (70 -> 90 us) while the MFMA/VALU-bound attention kernel is stable.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void rd(const uint4* __restrict__ p, size_t n_per_block, uint4* out, int reps) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  const uint4* q = p + (size_t)blockIdx.x * n_per_block;
  for (int r = 0; r < reps; ++r)
    for (size_t i = threadIdx.x; i < n_per_block; i += blockDim.x) { uint4 v = q[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  if (acc.x == 0x12345678) out[0] = acc;
}
// every block reads the SAME region (the chain kernels' weight stream pattern)
__global__ void rd_shared(const uint4* __restrict__ p, size_t n, uint4* out, int reps) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int r = 0; r < reps; ++r)
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) { uint4 v = p[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  if (acc.x == 0x12345678) out[0] = acc;
}
template <typename F> float time_it(F f, int iters = 5) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); f();
  CK(hipEventRecord(e0)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters;
}
int main() {
  uint4 *buf, *out; const size_t big = (size_t)2 << 30;
  CK(hipMalloc(&buf, big)); CK(hipMalloc(&out, 64)); CK(hipMemset(buf, 1, big));
  for (size_t total : {(size_t)2 << 30, (size_t)128 << 20, (size_t)16 << 20}) {
    const int blocks = 2048; const size_t per = total / 16 / blocks; const int reps = total >= ((size_t)1 << 30) ? 1 : 16;
    float ms = time_it([&] { rd<<<blocks, 256>>>(buf, per, out, reps); });
    printf("private read %6zu MiB x%2d: %8.1f GB/s\n", total >> 20, reps, (double)total * reps / ms * 1e-6);
  }
  for (size_t region : {(size_t)4 << 20, (size_t)1 << 20}) {
    float ms = time_it([&] { rd_shared<<<200, 256>>>(buf, region / 16, out, 4); });
    printf("200 blocks all reading the same %zu MiB x4: %8.1f GB/s aggregate, %6.1f GB/s per CU\n", region >> 20, 200.0 * region * 4 / ms * 1e-6,
           (double)region * 4 / ms * 1e-6);
  }
  return 0;
}
