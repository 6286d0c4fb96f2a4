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
// the sub-step kernel's launch shape with no drone flying: nf fat fill waves first (grid-stride over everything
// past the first `tiny` KB), then `tiny` one-wave workgroups that load one flag, wait for it, store 1 KB and retire
__global__ void mix(float4* dst, size_t n, int nf, int tiny, const int* flags, int mode) {
  const int w = blockIdx.x;
  const float4 one = make_float4(1, 1, 1, 1);
  if (w < nf) {
    for (size_t q = (size_t)tiny * 64 + (size_t)w * 64 + threadIdx.x; q < n; q += (size_t)nf * 64) dst[q] = one;
    return;
  }
  const int t = w - nf;
  int f = (mode & 1) ? flags[(size_t)t * 64 + threadIdx.x] : 0;
  if (mode & 2) dst[(size_t)t * 64 + threadIdx.x] = one;
  if (f == 12345) dst[0] = one;  // keeps the load alive
}
template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
// the same, with another buffer swept between two timed launches (what the engage/observe kernel and the state
// planes do to the caches between two sub-step launches): separates HBM from Infinity-Cache hits on a rewrite
__global__ void scrub(float4* b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = b[i]; v.x += 1.0f; b[i] = v; }
}
template <typename F> float timeit_scrubbed(F f, float4* other, size_t n_other) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float total = 0;
  for (int i = 0; i < 6; ++i) {
    scrub<<<4096, 256>>>(other, n_other);
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    if (i) total += ms;
  }
  return total / 5;
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
  for (int g : {128, 256, 512, 1024, 2048, 4096}) {  // a few fat single-wave workgroups (the sub-step kernel's fill waves)
    char name[64]; snprintf(name, sizeof name, "grid-stride dwordx4 (%dx64)", g);
    rep(name, timeit([&] { gs<float4><<<g, 64>>>((float4*)buf, bytes / 16); }));
  }
  int* flags; hipMalloc(&flags, 11264 * 64 * 4); hipMemset(flags, 0, 11264 * 64 * 4);
  for (int mode : {0, 1, 2, 3})
    for (int nf : {256, 512}) {
      char name[64]; snprintf(name, sizeof name, "mix nf=%d tiny=11264 mode=%d", nf, mode);
      rep(name, timeit([&] { mix<<<nf + 11264, 64>>>((float4*)buf, bytes / 16, nf, 11264, flags, mode); }));
    }
  char* other; hipMalloc(&other, 512ull << 20);
  for (size_t mb : {64, 128, 256, 512}) {
    char name[64]; snprintf(name, sizeof name, "grid-stride x4 (512x64), %zu MB scrubbed", mb);
    rep(name, timeit_scrubbed([&] { gs<float4><<<512, 64>>>((float4*)buf, bytes / 16); }, (float4*)other, (mb << 20) / 16));
    snprintf(name, sizeof name, "hipMemsetD32Async, %zu MB scrubbed", mb);
    rep(name, timeit_scrubbed([&] { hipMemsetD32Async((hipDeviceptr_t)buf, 0x3f800000, bytes / 4, 0); }, (float4*)other, (mb << 20) / 16));
  }
  rep("wave chunk dwordx4 K=24", timeit([&] { size_t n = bytes / 16; chunk<float4, 24><<<(n + 24 * 64 - 1) / (24 * 64), 64>>>((float4*)buf, n); }));
  rep("wave chunk dwordx2 K=48", timeit([&] { size_t n = bytes / 8; chunk<float2, 48><<<(n + 48 * 64 - 1) / (48 * 64), 64>>>((float2*)buf, n); }));
  rep("wave chunk dword   K=96", timeit([&] { size_t n = bytes / 4; chunk<float, 96><<<(n + 96 * 64 - 1) / (96 * 64), 64>>>((float*)buf, n); }));
  rep("wave chunk dword   K=32", timeit([&] { size_t n = bytes / 4; chunk<float, 32><<<(n + 32 * 64 - 1) / (32 * 64), 64>>>((float*)buf, n); }));
  return 0;
}
