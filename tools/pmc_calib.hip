// pmc_calib.hip — known-byte-count streaming kernels to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on
// gfx950 for the access widths the environment kernels use (MI355X_MICROARCH.md "HBM": FETCH_SIZE reads 1/2
// of a 16 B/lane stream; other widths are uncalibrated).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/pmc_calib tools/pmc_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib_f -o c -- ./gpurun_out/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read4(const float* __restrict__ src, float* __restrict__ out, size_t n) {  // dword loads, 256 B per wave
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
  if (acc == 123.456f) out[0] = acc;
}
__global__ void read16(const float4* __restrict__ src, float* __restrict__ out, size_t n4) {  // dwordx4 loads
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 123.456f) out[0] = acc;
}
__global__ void write4(float* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = 1.0f;
}
__global__ void write16(float4* __restrict__ dst, size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_float4(1, 1, 1, 1);
}
int main() {
  const size_t bytes = 1ull << 30;  // 1 GiB: far beyond the 256 MiB Infinity Cache
  float *a, *b;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes);
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    read4<<<4096, 256>>>(a, b, bytes / 4);
    read16<<<4096, 256>>>((const float4*)a, b, bytes / 16);
    write4<<<4096, 256>>>(b, bytes / 4);
    write16<<<4096, 256>>>((float4*)b, bytes / 16);
  }
  hipDeviceSynchronize();
  printf("each kernel moves %zu bytes\n", bytes);
  return 0;
}
