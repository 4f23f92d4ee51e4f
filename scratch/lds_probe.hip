// LDS bank-conflict probe (scratch; not product): cycles per wave-instruction for the access patterns of the chain kernels.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/lds_probe.hip -o scratch/lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
enum { R128 = 0, W128 = 1, W64 = 2, R64 = 3 };
template <int OP, int PAT>
__global__ void probe(unsigned long long* out, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
  const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
  int addr = 0;
  if (PAT == 0) addr = lane * 16;                                                   // linear 16 B
  if (PAT == 1) addr = l15 * 1024 + (((5 + g) ^ l15) << 4);                          // A fragment read: 16 rows of a 1 KiB-pitch panel, swizzled chunk
  if (PAT == 2) addr = l15 * 128 + ((g ^ ((l15 >> 1) & 7)) << 4);                    // W fragment read, 16-row slice (8 waves)
  if (PAT == 3) addr = (l15 + 16 * (g & 1)) * 128 + (((g >> 1) ^ (((l15 + 16 * (g & 1)) >> 1) & 7)) << 4);   // W fragment, 32-row slice (4 waves, j = g&1 mimic)
  if (PAT == 4) addr = l15 * 1024 + (((4 + g) ^ l15) << 4) + 8;                      // 8-byte panel write of the 8-wave shape (ln_write / GELU): half a chunk
  if (PAT == 5) addr = l15 * 1024 + (((4 + g) ^ l15) << 4);                          // 16-byte panel write of the 4-wave shape
  if (PAT == 6) addr = l15 * 64 + g * 8;                                             // pair staging write (8 waves): [16 rows][32 cols], 8 B
  if (PAT == 7) addr = (l15) * 96 + g * 8;                                           // V^T staging write: column pitch 96 B, 8 B
  if (PAT == 8) addr = l15 * 512 + (((4 + g) ^ l15) << 4) + 8;                       // 8-byte write into the 256 B... hidden chunk (pitch 256 B)
  if (PAT == 9) addr = l15 * 256 + (((4 + g) ^ l15) << 4) + 8;                       // 8-byte GELU write, hidden chunk pitch 256 B (HLD = 128)
  u32x4 v = {1, 2, 3, 4};
  unsigned acc = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 2048; ++i) {
    const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)(lds + addr);
    if (OP == R128) { u32x4 r; asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(a)); asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory"); acc ^= 1; asm volatile("" :: "v"(r)); }
    if (OP == R64) { u32x2 r; asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(a)); asm volatile("" :: "v"(r)); }
    if (OP == W128) asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(v) : "memory");
    if (OP == W64) { u32x2 w = {v[0], v[1]}; asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(w) : "memory"); }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[threadIdx.x >> 6] = t1 - t0;
  if (acc == 12345) sink[0] = acc;
}
template <int OP, int PAT>
int run(const char* name, int waves, unsigned long long* d, unsigned* sink) {
  probe<OP, PAT><<<1, 64 * waves>>>(d, sink);
  CK(hipDeviceSynchronize());
  unsigned long long h[8];
  CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  printf("  %-64s waves=%d  %6.1f cycles per wave-instruction\n", name, waves, (double)h[0] / 2048.0);
  return 0;
}
int main() {
  unsigned long long* d; unsigned* sink;
  CK(hipMalloc(&d, 64)); CK(hipMalloc(&sink, 64));
  for (int waves : {1, 4, 8}) {
    run<R128, 0>("read b128 linear", waves, d, sink);
    run<R128, 1>("read b128 A fragment (1 KiB pitch, swizzled)", waves, d, sink);
    run<R128, 2>("read b128 W fragment, 16-row slice (8-wave shape)", waves, d, sink);
    run<R128, 3>("read b128 W fragment, 32-row slice (4-wave shape)", waves, d, sink);
    run<W128, 0>("write b128 linear", waves, d, sink);
    run<W128, 5>("write b128 panel (4-wave ln_write / GELU)", waves, d, sink);
    run<W64, 4>("write b64 panel, 1 KiB pitch (8-wave ln_write)", waves, d, sink);
    run<W64, 9>("write b64 hidden chunk, 256 B pitch (8-wave GELU)", waves, d, sink);
    run<W64, 6>("write b64 pair staging [16][32 cols]", waves, d, sink);
    run<W64, 7>("write b64 V^T staging (96 B pitch)", waves, d, sink);
  }
  return 0;
}
