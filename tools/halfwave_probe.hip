// Does a wave64 with only its lower 32 (or 16) lanes active issue VALU instructions faster on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/halfwave_probe.hip -o ab/halfwave_probe && ab/halfwave_probe
// Launches enough single-wave workgroups to put 4 waves on every SIMD, each running a long dependent FMA chain with
// `active` lanes enabled, and reports the time per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void chain(float* out, int active, int iters) {
  if ((int)threadIdx.x >= active) return;
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
  for (int i = 0; i < iters; ++i) {
    a = fmaf(a, b, c); d = fmaf(d, b, a); c = fmaf(c, b, d); a = fmaf(a, d, c);
    a = fmaf(a, b, c); d = fmaf(d, b, a); c = fmaf(c, b, d); a = fmaf(a, d, c);
  }
  out[blockIdx.x * 64 + threadIdx.x] = a + c + d;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int active : {64, 48, 32, 16, 64}) {
    chain<<<4096, 64>>>(out, active, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) chain<<<4096, 64>>>(out, active, 20000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("active lanes %2d: %.3f ms per launch (4096 waves x 160000 dependent FMAs)\n", active, ms / 10);
  }
  return 0;
}
