// Does vmcnt retire a younger STORE before an older LOAD on gfx950?  (scratch; not product)
// One wave per block: [cold global load][store][s_waitcnt vmcnt(1)][copy the load's destination].  In-order retirement: the copy
// always holds the loaded value.  Out-of-order (store acknowledged first): the copy can still hold the sentinel.
// Second kernel: the same with the load as LDS-DMA (global_load_lds) and a ds_read after the counted wait.
// hipcc --offload-arch=gfx950 -O3 scratch/vmcnt_order_probe.hip -o scratch/vmcnt_order_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void fill_kernel(int* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (int)(i * 2654435761u) | 1;
}
__global__ __launch_bounds__(64) void probe_reg(const int* big, int* sink, int* out, size_t stride, int nstores) {
  const int lane = threadIdx.x;
  const size_t idx = ((size_t)blockIdx.x * 977 + 13) * stride + (size_t)lane * 32;   // 64 distinct cold 128-byte lines per wave
  const int* lp = big + idx;
  int* sp = sink + (size_t)blockIdx.x * 64 * 8 + lane;
  int v = 0x5e5e5e5e, copy = 0;
  if (nstores == 0)   // positive control: a wait that covers nothing -- the copy MUST be early
    asm volatile("global_load_dword %0, %2, off\n\tglobal_store_dword %3, %4, off\n\ts_waitcnt vmcnt(2)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                 : "+v"(v), "=&v"(copy) : "v"(lp), "v"(sp), "v"(lane) : "memory");
  else if (nstores == 1)
    asm volatile("global_load_dword %0, %2, off\n\tglobal_store_dword %3, %4, off\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                 : "+v"(v), "=&v"(copy) : "v"(lp), "v"(sp), "v"(lane) : "memory");
  else
    asm volatile("global_load_dword %0, %2, off\n\tglobal_store_dword %3, %4, off\n\tglobal_store_dword %3, %4, off offset:256\n\t"
                 "global_store_dword %3, %4, off offset:512\n\ts_waitcnt vmcnt(3)\n\tv_mov_b32 %1, %0\n\ts_waitcnt vmcnt(0)"
                 : "+v"(v), "=&v"(copy) : "v"(lp), "v"(sp), "v"(lane) : "memory");
  out[(size_t)blockIdx.x * 64 + lane] = (copy == v) ? 0 : 1;   // 1: the copy was taken before the load had landed
}
__global__ __launch_bounds__(64) void probe_lds(const int* big, int* sink, int* out, size_t stride) {
  __shared__ int lds[64 * 4];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) lds[i] = 0x5e5e5e5e;
  __syncthreads();
  const size_t idx = ((size_t)blockIdx.x * 977 + 13) * stride + (size_t)lane * 32;
  const int* lp = big + idx;
  int* sp = sink + (size_t)blockIdx.x * 64 * 8 + lane;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)lp, (__attribute__((address_space(3))) void*)lds, 4, 0, 0);
  int copy;
  asm volatile("global_store_dword %1, %2, off\n\tglobal_store_dword %1, %2, off offset:256\n\ts_waitcnt vmcnt(2)\n\t"
               "ds_read_b32 %0, %3\n\ts_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)"
               : "=&v"(copy) : "v"(sp), "v"(lane), "v"((unsigned)(uintptr_t)(__attribute__((address_space(3))) int*)(lds + lane)) : "memory");
  out[(size_t)blockIdx.x * 64 + lane] = (copy == 0x5e5e5e5e) ? 1 : 0;
}
int main() {
  const size_t n = (size_t)1 << 30;   // 4 GiB of ints: far beyond L2 + MALL
  int *big, *sink, *out;
  const int blocks = 4096;
  CK(hipMalloc(&big, n * 4)); CK(hipMalloc(&sink, (size_t)blocks * 64 * 8 * 4 + 4096)); CK(hipMalloc(&out, (size_t)blocks * 64 * 4));
  fill_kernel<<<4096, 256>>>(big, n);
  CK(hipDeviceSynchronize());
  const size_t stride = n / ((size_t)blocks * 977 + 64) ;
  std::vector<int> h((size_t)blocks * 64);
  for (int rep = 0; rep < 3; ++rep) {
    for (int ns : {0, 1, 3}) {
      probe_reg<<<blocks, 64>>>(big + rep * 4099, sink, out, stride, ns);
      CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
      long bad = 0; for (int x : h) bad += x;
      printf("rep %d  load -> %d store(s) -> vmcnt(%d)%s: %ld of %zu lanes copied BEFORE the older load landed\n", rep, ns ? ns : 1, ns ? ns : 2,
             ns ? "" : " [control: the wait covers nothing]", bad, h.size());
    }
    probe_lds<<<blocks, 64>>>(big + rep * 4099 + 2048, sink, out, stride);
    CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    long bad = 0; for (int x : h) bad += x;
    printf("rep %d  LDS-DMA -> 2 stores -> vmcnt(2) -> ds_read: %ld of %zu lanes read the OLD LDS bytes\n", rep, bad, h.size());
  }
  return 0;
}
