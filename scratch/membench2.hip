// scratch microbenchmark (round 3): K5's memory skeleton without its arithmetic.  One wave per chunk, 1 KiB per wave-instruction,
// four tile loads in flight, 3 workgroups of 8 waves per CU (50 KiB of LDS each) -- then one ingredient at a time: stores that start
// at an odd multiple of 16 bytes, the side-car stores, a VALU delay per tile, an LDS round trip per tile.  Which one costs the
// 20 % between the slab copy (5.4 TB/s) and K5 (4.4 TB/s)?   hipcc --offload-arch=gfx950 -O3 scratch/membench2.hip -o scratch/membench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int SHIFT, bool SIDE, int VALU, bool LDSRT, int NB>
__global__ __launch_bounds__(512, 6) void k(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t* __restrict__ side,
                                             uint64_t chunk_vec, uint32_t nchunks) {
  __shared__ uint32_t pad[12800];  // 50 KiB: three workgroups per CU, like k_emit
  __shared__ uint4 stage[8][64 + 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) pad[blockIdx.x & 1023] = 0;
  const uint32_t c = blockIdx.x * 8 + wave;
  if (c >= nchunks) return;
  const uint4* p = in + (uint64_t)c * chunk_vec + lane;
  uint4* q = out + (uint64_t)c * chunk_vec + lane + (SHIFT ? (c % 7) + 1 : 0);  // units of 16 B: an odd phase per chunk
  uint32_t* s = side + ((uint64_t)c * chunk_vec + lane) / 4;
  const uint64_t ntiles = chunk_vec / 64;
  uint4 buf[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) buf[j] = p[j * 64];
  __builtin_amdgcn_s_waitcnt(0x0F70);
  for (uint64_t it = 0; it + NB <= ntiles; it += NB) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      uint4 v = buf[j];
      __builtin_amdgcn_sched_barrier(0);
      const uint64_t nx = it + NB + j < ntiles ? it + NB + j : ntiles - 1;
      buf[j] = p[nx * 64];
      __builtin_amdgcn_sched_barrier(0);
      uint32_t a = v.x;
#pragma unroll
      for (int i = 0; i < VALU; ++i) a = a * 2654435761u + v.y;  // a dependent chain of VALU work
      v.x = a;
      if (LDSRT) {
        stage[wave][lane] = v;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        v = stage[wave][(lane + 1) & 63];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      if (SIDE) s[(it + j) * 16] = v.y;
      q[(it + j) * 64] = v;
    }
  }
}

template <int SHIFT, bool SIDE, int VALU, bool LDSRT, int NB>
int run(const char* name, const uint4* d, uint4* o, uint32_t* side, uint64_t n) {
  const uint32_t nchunks = 6144;
  const uint64_t chunk = ((n / nchunks + 16383) / 16384) * 16384;
  const uint32_t used = (uint32_t)(n / chunk);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<SHIFT, SIDE, VALU, LDSRT, NB>), dim3((used + 7) / 8), dim3(512), 0, 0, d, o, side, chunk / 16, used);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int R = 6;
  for (int r = 0; r < R; ++r) hipLaunchKernelGGL((k<SHIFT, SIDE, VALU, LDSRT, NB>), dim3((used + 7) / 8), dim3(512), 0, 0, d, o, side, chunk / 16, used);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= R;
  printf("%4llu MiB %-46s shift%d side%d valu%-3d lds%d nb%d  %.4f ms  %7.1f GB/s\n", (unsigned long long)(n >> 20), name, SHIFT, (int)SIDE, VALU, (int)LDSRT, NB, ms,
         2.0 * used * chunk / ms / 1e6);
  fflush(stdout);
  return 0;
}

int main() {
  uint4 *d, *o; uint32_t* side;
  const uint64_t cap = (4ull << 30) + (64 << 20);
  CK(hipMalloc(&d, cap)); CK(hipMalloc(&o, cap)); CK(hipMalloc(&side, cap / 16 + 4096)); CK(hipMemset(d, 1, cap)); CK(hipMemset(o, 2, cap));
  for (uint64_t n : {256ull << 20, 4ull << 30}) {
    run<0, false, 0, false, 4>("skeleton", d, o, side, n);
    run<0, false, 0, false, 2>("skeleton, 2 loads in flight", d, o, side, n);
    run<1, false, 0, false, 4>("+ stores at an odd 16-byte phase", d, o, side, n);
    run<1, true, 0, false, 4>("+ side-car stores", d, o, side, n);
    run<1, true, 64, false, 4>("+ 64 dependent VALU per tile", d, o, side, n);
    run<1, true, 160, false, 4>("+ 160 dependent VALU per tile", d, o, side, n);
    run<1, true, 0, true, 4>("+ LDS round trip per tile (no VALU)", d, o, side, n);
    run<1, true, 160, true, 4>("+ 160 VALU + LDS round trip", d, o, side, n);
    run<0, false, 160, true, 4>("aligned, no side-car, 160 VALU + LDS", d, o, side, n);
  }
  return 0;
}
