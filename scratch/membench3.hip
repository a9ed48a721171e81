// scratch/membench3.hip -- K7's memory skeleton with a variable number of groups in flight per wave (round 4's question:
// would "fewer waves, two groups of loads in flight each" pay?).  No decode: a wave takes groups of 4096 output bytes whose
// compressed span is 4128 bytes (uniform bytes), loads the span into registers (5 x 16 B per lane, D groups ahead), stages
// it through its LDS tile, reads 64 bytes per lane back (the transposition) and stores 4 x 1 KiB.  W waves per CU.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/membench3 scratch/membench3.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 ldnt(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
__device__ __forceinline__ void stnt(void* p, u32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); }

// MODE 0: spans at a pitch of 4128 bytes (128-byte lines shared by two spans at most at the ends); 1: at the pitch uniform
// bytes really have (4118.5 bytes: starts anywhere, rounded down to 16); 2: ... and the start comes from a side-car word that
// is loaded one group further ahead (the dependent chain side-car -> span of the real kernel)
template <int W, int D, int MODE>
__global__ __launch_bounds__(W * 64) void k_skel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t ngroups, const uint64_t* __restrict__ meta) {
  __shared__ u32x4 tile[W][5 * 64 + 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t nw = gridDim.x * W, wid = blockIdx.x * W + wave;
  u32x4 R[D][5];
  uint32_t g[D + 1];
#pragma unroll
  for (int d = 0; d <= D; ++d) g[d] = wid + d * nw;
  uint64_t m_next = 0;  // side-car word of the group whose span is issued next
  auto issue = [&](int d, uint32_t grp) {
    const uint32_t gc = grp < ngroups ? grp : ngroups - 1;
    uint64_t off = MODE == 0 ? (uint64_t)gc * 4128 : (((uint64_t)gc * 32948) >> 3) & ~15ull;
    if (MODE == 2) {
      off = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m_next) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(m_next >> 32)) << 32);
      if (off > (uint64_t)ngroups * 4128) off = 0;  // (never: a belt for a microbenchmark that must not fault)
      const uint32_t gn = grp + nw < ngroups ? grp + nw : ngroups - 1;
      m_next = meta[gn];
    }
    const uint8_t* p = in + off;
#pragma unroll
    for (int k = 0; k < 5; ++k) R[d][k] = ldnt(p + (k * 1024 + lane * 16 < 4128 ? k * 1024 + lane * 16 : 0));
  };
  if (MODE == 2) m_next = meta[g[0] < ngroups ? g[0] : ngroups - 1];
#pragma unroll
  for (int d = 0; d < D; ++d) issue(d, g[d]);
  while (g[0] < ngroups) {
    // stage the oldest group, re-issue its registers for the group D ahead
#pragma unroll
    for (int k = 0; k < 5; ++k) tile[wave][k * 64 + lane] = R[0][k];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t cur = g[0];
#pragma unroll
    for (int d = 0; d + 1 < D; ++d) {
#pragma unroll
      for (int k = 0; k < 5; ++k) R[d][k] = R[d + 1][k];
    }
#pragma unroll
    for (int d = 0; d < D; ++d) g[d] = g[d + 1];
    issue(D - 1, g[D - 1]);
    // "decode": 64 bytes per lane out of the tile, then the coalesced copy-out
    u32x4 o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = tile[wave][(lane * 4 + q) % (5 * 64)];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint8_t* og = out + (uint64_t)cur * 4096 + lane * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) stnt(og + q * 1024, o[q]);
    g[D] = g[D - 1] + nw;  // (static stride: one global ticket word saturates at ~88 tickets/us -- 0.75 ms for 65536 groups)
  }
}
template <int W, int D, int MODE> void run(const uint8_t* in, uint8_t* out, uint64_t n, const uint64_t* meta) {
  const uint32_t ngroups = (uint32_t)(n / 4096);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 10;
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) {
      hipLaunchKernelGGL((k_skel<W, D, MODE>), dim3(256), dim3(W * 64), 0, 0, in, out, ngroups, meta);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms / reps < best) best = ms / reps;
  }
  printf("%5llu MiB  %2d waves/CU  %d group(s) in flight  mode %d   %.4f ms   %.0f GB/s (read+write)\n", (unsigned long long)(n >> 20), W, D, MODE, best,
         (double)ngroups * (4128 + 4096) / best / 1e6);
}
__global__ void k_meta(uint64_t* m, uint32_t n) { for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m[i] = (((uint64_t)i * 32948) >> 3) & ~15ull; }
int main() {
  for (uint64_t mib : {256ull, 4096ull}) {
    const uint64_t n = mib << 20;
    uint8_t *in, *out; uint64_t* meta;
    CK(hipMalloc(&in, n / 4096 * 4128 + 8192)); CK(hipMalloc(&out, n)); CK(hipMalloc(&meta, n / 4096 * 8 + 64));
    CK(hipMemset(in, 1, n / 4096 * 4128 + 8192));
    hipLaunchKernelGGL(k_meta, dim3(256), dim3(256), 0, 0, meta, (uint32_t)(n / 4096));
    CK(hipDeviceSynchronize());
    run<16, 1, 0>(in, out, n, meta);
    run<16, 2, 0>(in, out, n, meta);
    run<8, 1, 0>(in, out, n, meta);
    run<8, 2, 0>(in, out, n, meta);
    run<16, 1, 1>(in, out, n, meta);
    run<16, 1, 2>(in, out, n, meta);
    run<16, 2, 2>(in, out, n, meta);
    CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(meta));
  }
}
