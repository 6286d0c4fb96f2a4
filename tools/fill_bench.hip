// fill_bench.hip — the sub-step kernel's background fill in isolation: W one-wave workgroups, scalar loop of non-temporal 16-byte buffer
// stores, for the two buffer sizes of the bench (266 MB: stage03 x 65 536 envs; 1.6 GB: level5) and three block shapes:
//   strided : wave w writes 1 KB blocks w, w + W, w + 2 W, ...                     (what substeps_kernel does)
//   run4    : ... 4 KB runs (4 stores with immediate offsets), run index strided by W
//   slab    : wave w owns one contiguous slab of total / W bytes
// hipcc --offload-arch=gfx950 -O3 -o gpurun_out/fill_bench tools/fill_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
template <int MODE> __global__ __launch_bounds__(64) void fill(float* base, uint32_t n_blocks, uint32_t W) {
  const uint32_t wave = blockIdx.x;
  const int voff = threadIdx.x * 16;
  const u4 ones = {0x3f800000u, 0x3f800000u, 0x3f800000u, 0x3f800000u};
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(n_blocks << 10), 0x00020000);
  if (MODE == 0) {
    for (uint32_t b = wave; b < n_blocks; b += W) __builtin_amdgcn_raw_buffer_store_b128(ones, rs, voff, (int)(b << 10), 2);
  } else if (MODE == 1) {
    for (uint32_t r = wave; r * 4 + 3 < n_blocks; r += W) {
      const int so = (int)(r << 12);
      __builtin_amdgcn_raw_buffer_store_b128(ones, rs, voff, so, 2);
      __builtin_amdgcn_raw_buffer_store_b128(ones, rs, voff + 1024, so, 2);
      __builtin_amdgcn_raw_buffer_store_b128(ones, rs, voff + 2048, so, 2);
      __builtin_amdgcn_raw_buffer_store_b128(ones, rs, voff + 3072, so, 2);
    }
  } else {
    const uint32_t per = n_blocks / W;
    for (uint32_t b = wave * per; b < (wave + 1) * per; ++b) __builtin_amdgcn_raw_buffer_store_b128(ones, rs, voff, (int)(b << 10), 2);
  }
}
int main() {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (size_t bytes : {(size_t)266 << 20, (size_t)1596 << 20}) {
    float* buf; hipMalloc(&buf, bytes);
    const uint32_t n_blocks = (uint32_t)(bytes >> 10);
    for (int mode = 0; mode < 3; ++mode)
      for (uint32_t W : {64u, 128u, 256u, 512u, 1024u, 2048u, 4096u}) {
        float tot = 0;
        for (int it = 0; it < 6; ++it) {
          hipEventRecord(a);
          if (mode == 0) fill<0><<<W, 64>>>(buf, n_blocks, W); else if (mode == 1) fill<1><<<W, 64>>>(buf, n_blocks, W); else fill<2><<<W, 64>>>(buf, n_blocks, W);
          hipEventRecord(b); hipEventSynchronize(b);
          float ms; hipEventElapsedTime(&ms, a, b);
          if (it) tot += ms;
        }
        printf("%5zu MB  %-7s W = %4u : %7.1f us  %5.2f TB/s\n", bytes >> 20, mode == 0 ? "strided" : mode == 1 ? "run4" : "slab", W, tot / 5 * 1e3, bytes / (tot / 5 * 1e-3) / 1e12);
      }
    hipFree(buf);
  }
  return 0;
}
