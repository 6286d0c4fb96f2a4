// stream_bench.hip — which store shape streams constants to HBM fastest on gfx950?  (choice of the LIDAR
// background store in the sub-step kernel).  hipcc --offload-arch=gfx950 -O3 -o gpurun_out/stream_bench tools/stream_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T> __device__ T ones();
template <> __device__ float ones<float>() { return 1.0f; }
template <> __device__ float2 ones<float2>() { return make_float2(1, 1); }
template <> __device__ float4 ones<float4>() { return make_float4(1, 1, 1, 1); }
// grid-stride: consecutive lanes consecutive elements, whole grid sweeps the buffer
template <typename T> __global__ void gs(T* dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = ones<T>();
}
// per-wave contiguous chunk of K*64 elements (the sub-step kernel's shape), one wave per block
template <typename T, int K> __global__ void chunk(T* dst, size_t n) {
  size_t q0 = (size_t)blockIdx.x * (K * 64) + threadIdx.x;
#pragma unroll
  for (int k = 0; k < K; ++k) { size_t i = q0 + (size_t)k * 64; if (i < n) dst[i] = ones<T>(); }
}
template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
  const size_t bytes = 266ull << 20;  // the 65 536-env LIDAR background
  char* buf; hipMalloc(&buf, bytes);
  auto rep = [&](const char* name, float ms) { printf("%-34s %7.1f us  %5.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12); };
  rep("hipMemsetD32Async", timeit([&] { hipMemsetD32Async((hipDeviceptr_t)buf, 0x3f800000, bytes / 4, 0); }));
  rep("grid-stride dword  (4096x256)", timeit([&] { gs<float><<<4096, 256>>>((float*)buf, bytes / 4); }));
  rep("grid-stride dwordx2(4096x256)", timeit([&] { gs<float2><<<4096, 256>>>((float2*)buf, bytes / 8); }));
  rep("grid-stride dwordx4(4096x256)", timeit([&] { gs<float4><<<4096, 256>>>((float4*)buf, bytes / 16); }));
  rep("grid-stride dwordx4(1024x256)", timeit([&] { gs<float4><<<1024, 256>>>((float4*)buf, bytes / 16); }));
  rep("grid-stride dwordx4(16384x256)", timeit([&] { gs<float4><<<16384, 256>>>((float4*)buf, bytes / 16); }));
  rep("wave chunk dwordx4 K=24", timeit([&] { size_t n = bytes / 16; chunk<float4, 24><<<(n + 24 * 64 - 1) / (24 * 64), 64>>>((float4*)buf, n); }));
  rep("wave chunk dwordx2 K=48", timeit([&] { size_t n = bytes / 8; chunk<float2, 48><<<(n + 48 * 64 - 1) / (48 * 64), 64>>>((float2*)buf, n); }));
  rep("wave chunk dword   K=96", timeit([&] { size_t n = bytes / 4; chunk<float, 96><<<(n + 96 * 64 - 1) / (96 * 64), 64>>>((float*)buf, n); }));
  rep("wave chunk dword   K=32", timeit([&] { size_t n = bytes / 4; chunk<float, 32><<<(n + 32 * 64 - 1) / (32 * 64), 64>>>((float*)buf, n); }));
  return 0;
}
