// scratch microbenchmark: does "one wave streams through its own chunk, 1 KiB at a time" cost HBM efficiency
// against a grid-stride copy?  (the access pattern of K5/K7)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0: grid-stride copy (every wave-instruction 1 KiB, consecutive waves consecutive KiB)
// MODE 1: wave w copies chunk w (chunk_bytes), 1 KiB per step, 2 loads in flight
// MODE 2: like 1 but TILE KiB per step (TILE loads in flight)
template <int MODE, int TILE>
__global__ __launch_bounds__(512) void k(const uint4* __restrict__ in, uint4* __restrict__ out, uint64_t nvec, uint32_t chunk_vec) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  if (MODE == 0) {
    for (uint64_t i = wave * 64 + lane; i < nvec; i += nwaves * 64) out[i] = in[i];
  } else {
    for (uint64_t c = wave; c * chunk_vec < nvec; c += nwaves) {
      const uint4* p = in + c * chunk_vec + lane;
      uint4* q = out + c * chunk_vec + lane;
      for (uint32_t i = 0; i < chunk_vec; i += 64 * TILE) {
        uint4 v[TILE];
#pragma unroll
        for (int t = 0; t < TILE; ++t) v[t] = p[i + 64 * t];
#pragma unroll
        for (int t = 0; t < TILE; ++t) q[i + 64 * t] = v[t];
      }
    }
  }
}

template <int MODE, int TILE>
int run(const char* name, const uint4* d, uint4* o, uint64_t n, uint32_t chunk_bytes, int grid) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE, TILE>), dim3(grid), dim3(512), 0, 0, d, o, n / 16, chunk_bytes / 16);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int R = 5;
  for (int r = 0; r < R; ++r) hipLaunchKernelGGL((k<MODE, TILE>), dim3(grid), dim3(512), 0, 0, d, o, n / 16, chunk_bytes / 16);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= R;
  printf("%-44s grid %5d chunk %4u KiB  %.3f ms  %.1f GB/s (read+write)\n", name, grid, chunk_bytes >> 10, ms, 2.0 * n / ms / 1e6);
  return 0;
}

int main() {
  const uint64_t n = 4ull << 30;
  uint4 *d, *o;
  CK(hipMalloc(&d, n)); CK(hipMalloc(&o, n)); CK(hipMemset(d, 1, n));
  run<0, 1>("grid-stride copy", d, o, n, 0, 2048);
  run<0, 1>("grid-stride copy", d, o, n, 0, 512);
  for (int g : {512, 768, 1024}) {
    run<1, 1>("per-wave chunk, 1 KiB steps", d, o, n, 512 << 10, g);
    run<1, 2>("per-wave chunk, 2 KiB steps", d, o, n, 512 << 10, g);
    run<1, 4>("per-wave chunk, 4 KiB steps", d, o, n, 512 << 10, g);
  }
  run<1, 2>("per-wave chunk, 2 KiB steps", d, o, n, 64 << 10, 768);
  run<1, 2>("per-wave chunk, 2 KiB steps", d, o, n, 32 << 10, 768);
  return 0;
}
