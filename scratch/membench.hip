// scratch microbenchmark (round 3): what read-only, write-only and read+write streams reach on this MI355X with the access
// shapes the codec kernels could use.  hipcc --offload-arch=gfx950 -O3 scratch/membench.hip -o scratch/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// MODE 0 copy, 1 read-only (sum -> one store per thread at the end), 2 write-only
// LAYOUT 0: grid-stride (consecutive workgroups touch consecutive 4 KiB); 1: every workgroup owns one contiguous slab
template <int MODE, int U, bool NTL, bool NTS, int LAYOUT>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ in, u32x4* __restrict__ out, uint64_t nvec) {
  const uint64_t T = (uint64_t)gridDim.x * 256;
  uint64_t i, step, end;
  if (LAYOUT == 0) { i = (uint64_t)blockIdx.x * 256 + threadIdx.x; step = T; end = nvec; }
  else { const uint64_t per = (nvec + gridDim.x - 1) / gridDim.x; i = blockIdx.x * per + threadIdx.x; step = 256; end = (blockIdx.x + 1) * per < nvec ? (blockIdx.x + 1) * per : nvec; }
  u32x4 acc = {0, 0, 0, 0};
  for (; i + (U - 1) * step < end; i += U * step) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = MODE == 2 ? (u32x4){(uint32_t)i, 1, 2, 3} : ld<NTL>(in + i + u * step);
#pragma unroll
    for (int u = 0; u < U; ++u) { if (MODE == 1) acc += v[u]; else st<NTS>(out + i + u * step, v[u]); }
  }
  for (; i < end; i += step) { u32x4 v = MODE == 2 ? (u32x4){1, 1, 2, 3} : ld<NTL>(in + i); if (MODE == 1) acc += v; else st<NTS>(out + i, v); }
  if (MODE == 1 && acc.x == 0x12345678u) out[threadIdx.x] = acc;
}

template <int MODE, int U, bool NTL, bool NTS, int LAYOUT>
int run(const char* name, const u32x4* d, u32x4* o, uint64_t n, int grid) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE, U, NTL, NTS, LAYOUT>), dim3(grid), dim3(256), 0, 0, d, o, n / 16);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int R = 6;
  for (int r = 0; r < R; ++r) hipLaunchKernelGGL((k<MODE, U, NTL, NTS, LAYOUT>), dim3(grid), dim3(256), 0, 0, d, o, n / 16);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= R;
  const double bytes = (MODE == 0 ? 2.0 : 1.0) * n;
  printf("%4llu MiB %-34s U%d ntl%d nts%d layout%d grid %5d  %.4f ms  %7.1f GB/s\n", (unsigned long long)(n >> 20), name, U, (int)NTL, (int)NTS, LAYOUT, grid, ms, bytes / ms / 1e6);
  fflush(stdout);
  return 0;
}

int main() {
  u32x4 *d, *o;
  const uint64_t cap = 4ull << 30;
  CK(hipMalloc(&d, cap)); CK(hipMalloc(&o, cap)); CK(hipMemset(d, 1, cap)); CK(hipMemset(o, 2, cap));
  for (uint64_t n : {256ull << 20, 4ull << 30}) {
    for (int g : {1024, 2048, 4096}) {
      run<1, 4, true, true, 0>("read-only", d, o, n, g);
      run<2, 4, true, true, 0>("write-only nt", d, o, n, g);
      run<2, 4, true, false, 0>("write-only plain", d, o, n, g);
      run<0, 4, true, true, 0>("copy nt/nt", d, o, n, g);
      run<0, 4, true, false, 0>("copy ntload/plain store", d, o, n, g);
      run<0, 4, false, false, 0>("copy plain/plain", d, o, n, g);
      run<0, 8, true, true, 0>("copy nt/nt", d, o, n, g);
      run<0, 2, true, true, 0>("copy nt/nt", d, o, n, g);
      run<0, 4, true, true, 1>("copy nt/nt slab", d, o, n, g);
      run<0, 8, false, false, 1>("copy plain slab", d, o, n, g);
    }
    run<0, 4, true, true, 0>("copy nt/nt", d, o, n, 8192);
    run<0, 4, true, true, 0>("copy nt/nt", d, o, n, 512);
    run<0, 8, true, true, 0>("copy nt/nt", d, o, n, 512);
    run<0, 8, true, true, 0>("copy nt/nt", d, o, n, 256);
  }
  return 0;
}
