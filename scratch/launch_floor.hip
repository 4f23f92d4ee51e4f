// Dispatch floor of dependent launches: stream launches vs a captured graph (scratch; not product).
// hipcc --offload-arch=gfx950 -O3 scratch/launch_floor.hip -o scratch/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void empty_kernel(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }
__global__ void touch_kernel(float* p, int n) {   // read-modify-write n floats per launch (dependent chain through memory)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += 1.f;
}
int main() {
  float* buf; CK(hipMalloc(&buf, 16 << 20)); CK(hipMemset(buf, 0, 16 << 20));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 2000;
  for (int mode = 0; mode < 2; ++mode) {
    const int n = mode ? (1 << 18) : 0;   // 1 MiB touched per launch in mode 1
    auto launch = [&](hipStream_t st) {
      if (mode == 0) empty_kernel<<<1, 64, 0, st>>>(buf);
      else touch_kernel<<<n / 256, 256, 0, st>>>(buf, n);
    };
    for (int i = 0; i < 100; ++i) launch(s);
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < N; ++i) launch(s);
    CK(hipEventRecord(e1, s));
    auto t1 = std::chrono::steady_clock::now();
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mode %d stream launches: %.2f us per launch on the GPU, %.2f us per launch to enqueue on the host\n", mode, ms * 1e3 / N,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    // graph of 100 launches
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 100; ++i) launch(s);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mode %d graph of 100:    %.2f us per kernel node\n", mode, ms * 1e3 / 2000);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
