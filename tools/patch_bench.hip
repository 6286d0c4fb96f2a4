// patch_bench.hip — what does it cost to patch a few cells per env into the freshly streamed LIDAR background, and does the store
// shape matter?  (choice of the hit-cell store in engage_kernel.)  hipcc --offload-arch=gfx950 -O3 -o gpurun_out/patch_bench tools/patch_bench.hip
// One wave per 64 envs (lane = env), H hits per env, 3 channels per hit, cells pseudo-random per env:
//   dword    : 3 x 4-byte stores per hit (what the kernels do today)
//   quad     : 3 x 16-byte stores per hit: the 16-byte aligned group of the cell, ones elsewhere
//   seg64    : 3 x 64-byte stores per hit (4 dwordx4 of one lane): the 64-byte aligned segment of the cell, ones elsewhere
// each after (a) a non-temporal fill of the whole buffer (the sub-step kernel's fill waves), (b) a cached fill.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int TILE = 1014, CELLS = 338;
__global__ void fill(f4* dst, size_t n, int nt) {
  const f4 one = {1.f, 1.f, 1.f, 1.f};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (nt) __builtin_nontemporal_store(one, dst + i); else dst[i] = one;
  }
}
__device__ unsigned hashu(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int MODE> __global__ __launch_bounds__(64) void patch(float* lidar, int N, int H) {
  const int env = blockIdx.x * 64 + threadIdx.x;
  if (env >= N) return;
  for (int h = 0; h < H; ++h) {
    const int cell = hashu(env * 16 + h) % CELLS;
    const float r = 0.25f + 0.001f * h;
    for (int ch = 0; ch < 3; ++ch) {
      const size_t F = (size_t)env * TILE + ch * CELLS + cell;
      const float v = ch == 0 ? r : ch == 1 ? 0.2f : 0.1f;
      if (MODE == 0) lidar[F] = v;
      else if (MODE == 1) { f4 q = {1.f, 1.f, 1.f, 1.f}; q[F & 3] = v; *reinterpret_cast<f4*>(lidar + (F & ~(size_t)3)) = q; }
      else {
        f4* seg = reinterpret_cast<f4*>(lidar + (F & ~(size_t)15));
#pragma unroll
        for (int k = 0; k < 4; ++k) { f4 q = {1.f, 1.f, 1.f, 1.f}; if ((int)((F >> 2) & 3) == k) q[F & 3] = v; seg[k] = q; }
      }
    }
  }
}
// one thread per (env, hit): H times more waves in flight than `patch`
__global__ __launch_bounds__(256) void patch_par(float* lidar, int N, int H) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= N * H) return;
  const int env = t % N, h = t / N;
  const int cell = hashu(env * 16 + h) % CELLS;
  const float r = 0.25f + 0.001f * h;
  for (int ch = 0; ch < 3; ++ch) lidar[(size_t)env * TILE + ch * CELLS + cell] = ch == 0 ? r : ch == 1 ? 0.2f : 0.1f;
}
// one thread per (env, hit, channel), ENV-major: the 3 H stores of an env leave in the same store instruction (same 4 KB tile, same DRAM pages)
__global__ __launch_bounds__(256) void patch_envmajor(float* lidar, int N, int H) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= N * H * 3) return;
  const int env = t / (H * 3), r = t - env * H * 3, h = r / 3, ch = r - h * 3;
  const int cell = hashu(env * 16 + h) % CELLS;
  lidar[(size_t)env * TILE + ch * CELLS + cell] = ch == 0 ? 0.25f + 0.001f * h : ch == 1 ? 0.2f : 0.1f;
}
// full 128-byte lines: half a wave (32 lanes) writes the aligned line of one patch, ones elsewhere (NOT exact when two hits share a
// line: traffic experiment only)
__global__ __launch_bounds__(64) void patch_line(float* lidar, int N, int H) {
  const int lane = threadIdx.x, half = lane >> 5, l32 = lane & 31;
  for (int e = 0; e < 64; e += 2) {
    const int env = blockIdx.x * 64 + e + half;
    for (int h = 0; h < H; ++h) {
      const int cell = hashu(env * 16 + h) % CELLS;
      for (int ch = 0; ch < 3; ++ch) {
        const size_t F = (size_t)env * TILE + ch * CELLS + cell;
        const size_t base = F & ~(size_t)31;
        lidar[base + l32] = (base + l32 == F) ? 0.25f : 1.0f;
      }
    }
  }
}
// W lanes per patch write the aligned W*4-byte sector of the cell in ONE store instruction (64 / W patches per instruction),
// one thread group per (env, hit, channel); ones elsewhere (traffic experiment: not exact when two hits share a sector)
template <int W> __global__ __launch_bounds__(256) void patch_sector(float* lidar, int N, int H) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t item = t / W; const int l = (int)(t % W);
  if (item >= (size_t)N * H * 3) return;
  const int ch = (int)(item % 3); const size_t eh = item / 3;
  const int env = (int)(eh % N), h = (int)(eh / N);
  const int cell = hashu(env * 16 + h) % CELLS;
  const size_t F = (size_t)env * TILE + ch * CELLS + cell;
  const size_t base = F & ~(size_t)(W - 1);
  lidar[base + l] = (base + l == F) ? 0.25f : 1.0f;
}
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 65536;
  const size_t floats = (size_t)N * TILE;
  float* buf; hipMalloc(&buf, floats * 4 + 256);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int nt = 1; nt >= 0; --nt)
    for (int H : {2, 3, 6, 10})
      for (int mode = 0; mode < 9; ++mode) {
        float tot = 0;
        for (int it = 0; it < 6; ++it) {
          fill<<<2048, 256>>>((f4*)buf, floats / 4, nt);
          hipEventRecord(a);
          if (mode == 0) patch<0><<<N / 64, 64>>>(buf, N, H);
          else if (mode == 1) patch<1><<<N / 64, 64>>>(buf, N, H);
          else if (mode == 2) patch<2><<<N / 64, 64>>>(buf, N, H);
          else if (mode == 3) patch_par<<<(N * H + 255) / 256, 256>>>(buf, N, H);
          else if (mode == 4) patch_line<<<N / 64, 64>>>(buf, N, H);
          else if (mode == 5) patch_sector<8><<<(unsigned)(((size_t)N * H * 3 * 8 + 255) / 256), 256>>>(buf, N, H);
          else if (mode == 6) patch_sector<16><<<(unsigned)(((size_t)N * H * 3 * 16 + 255) / 256), 256>>>(buf, N, H);
          else if (mode == 8) patch_envmajor<<<(N * H * 3 + 255) / 256, 256>>>(buf, N, H);
          else patch_sector<32><<<(unsigned)(((size_t)N * H * 3 * 32 + 255) / 256), 256>>>(buf, N, H);
          hipEventRecord(b); hipEventSynchronize(b);
          float ms; hipEventElapsedTime(&ms, a, b);
          if (it) tot += ms;
        }
        printf("%s fill, %2d hits/env, %-6s: %6.1f us\n", nt ? "nontemporal" : "cached     ", H, mode == 0 ? "dword" : mode == 1 ? "quad" : mode == 2 ? "seg64" : mode == 3 ? "par" : mode == 4 ? "line128" : mode == 5 ? "sect32" : mode == 6 ? "sect64" : mode == 7 ? "sect128" : "envmaj", tot / 5 * 1e3);
      }
  return 0;
}
