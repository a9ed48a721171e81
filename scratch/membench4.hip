// scratch/membench4.hip (round 4): which GEOMETRY of K5's memory skeleton reaches the slab copy's rate?  Same ingredients as
// membench2's heaviest line (stores at an odd 16-byte phase, side-car stores, 160 dependent VALU + an LDS round trip per
// tile; 3 workgroups of 8 waves per CU), three ways to walk the input:
//   0  one wave per chunk, a tile re-loaded as soon as it is consumed (what k_emit does)
//   1  one wave per chunk, four tiles loaded TOGETHER every four tiles (two register sets)
//   2  a workgroup's eight waves sweep ONE chunk together: wave w takes tiles w, w + 8, w + 16 ... (8 KiB per round)
//   3  as 2, but the workgroup's chunk is swept in rounds of 4 tiles per wave: wave w takes tiles 4 w .. 4 w + 3 of every 32
//   hipcc --offload-arch=gfx950 -O3 scratch/membench4.hip -o scratch/membench4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int GEO, bool HEAVY>
__global__ __launch_bounds__(512, 6) void k(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t* __restrict__ side,
                                             uint64_t chunk_vec, uint32_t nchunks) {
  __shared__ uint32_t pad[12800];  // 50 KiB: three workgroups per CU, like k_emit
  __shared__ uint4 stage[8][64 + 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) pad[blockIdx.x & 1023] = 0;
  // GEO 0/1: chunk c = blockIdx * 8 + wave of chunk_vec vectors.  GEO 2/3: the workgroup owns 8 consecutive chunks as one.
  const uint64_t ntiles = chunk_vec / 64;  // per wave
  uint32_t c = blockIdx.x * 8 + wave;
  if (c >= nchunks) return;
  const uint4* p;
  uint4* q;
  uint32_t* s;
  auto tile_of = [&](uint64_t t) -> uint64_t {  // vector index of this wave's t-th tile, relative to the WORKGROUP's first vector
    if (GEO <= 1 || GEO >= 12) return (uint64_t)wave * chunk_vec + t * 64;
    if (GEO == 2) return (t * 8 + wave) * 64;
    return ((t >> 2) * 32 + wave * 4 + (t & 3)) * 64;
  };
  const uint64_t wg0 = (uint64_t)blockIdx.x * 8 * chunk_vec;
  p = in + wg0 + lane;
  q = out + wg0 + lane + ((blockIdx.x % 7) + 1);
  s = side + (wg0 + lane) / 4;
  auto work = [&](uint4 v, uint64_t t) {
    if (HEAVY) {
      uint32_t a = v.x;
#pragma unroll
      for (int i = 0; i < 160; ++i) a = a * 2654435761u + v.y;
      v.x = a;
      stage[wave][lane] = v;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      v = stage[wave][(lane + 1) & 63];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      s[tile_of(t) / 4] = v.y;
    }
    q[tile_of(t)] = v;
  };
  if constexpr (GEO == 1 || GEO >= 12) {
    constexpr int NS = GEO == 1 ? 4 : GEO - 10;
    uint4 A[NS], B[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) A[j] = p[tile_of(j)];
    for (uint64_t it = 0; it + 2 * NS <= ntiles; it += 2 * NS) {
#pragma unroll
      for (int j = 0; j < NS; ++j) B[j] = p[tile_of(it + NS + j)];
#pragma unroll
      for (int j = 0; j < NS; ++j) work(A[j], it + j);
#pragma unroll
      for (int j = 0; j < NS; ++j) A[j] = p[tile_of(it + 2 * NS + j < ntiles ? it + 2 * NS + j : ntiles - 1)];
#pragma unroll
      for (int j = 0; j < NS; ++j) work(B[j], it + NS + j);
    }
  } else {
    uint4 buf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) buf[j] = p[tile_of(j)];
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (uint64_t it = 0; it + 4 <= ntiles; it += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint4 v = buf[j];
        __builtin_amdgcn_sched_barrier(0);
        const uint64_t nx = it + 4 + j < ntiles ? it + 4 + j : ntiles - 1;
        buf[j] = p[tile_of(nx)];
        __builtin_amdgcn_sched_barrier(0);
        work(v, it + j);
      }
    }
  }
}

template <int GEO, bool HEAVY>
int run(const char* name, const uint4* d, uint4* o, uint32_t* side, uint64_t n) {
  const uint32_t nchunks = 6144;
  const uint64_t chunk = ((n / nchunks + 16383) / 16384) * 16384;
  const uint32_t used = (uint32_t)(n / chunk) / 8 * 8;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<GEO, HEAVY>), dim3(used / 8), dim3(512), 0, 0, d, o, side, chunk / 16, used);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int R = 6;
  for (int r = 0; r < R; ++r) hipLaunchKernelGGL((k<GEO, HEAVY>), dim3(used / 8), dim3(512), 0, 0, d, o, side, chunk / 16, used);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= R;
  printf("%4llu MiB  geometry %d %-52s %s  %.4f ms  %7.1f GB/s\n", (unsigned long long)(n >> 20), GEO, name, HEAVY ? "K5's work per tile" : "bare               ", ms,
         2.0 * used * chunk / ms / 1e6);
  fflush(stdout);
  return 0;
}

int main() {
  uint4 *d, *o; uint32_t* side;
  const uint64_t cap = (4ull << 30) + (64 << 20);
  CK(hipMalloc(&d, cap)); CK(hipMalloc(&o, cap)); CK(hipMalloc(&side, cap / 16 + 4096)); CK(hipMemset(d, 1, cap)); CK(hipMemset(o, 2, cap));
  for (uint64_t n : {256ull << 20, 4ull << 30}) {
    run<0, true>("wave per chunk, rolling re-load (4 in flight)", d, o, side, n);
    run<12, true>("wave per chunk, 2 tiles loaded together (2 sets)", d, o, side, n);
    run<13, true>("wave per chunk, 3 tiles loaded together (2 sets)", d, o, side, n);
    run<1, true>("wave per chunk, 4 tiles loaded together (2 sets)", d, o, side, n);
    run<16, true>("wave per chunk, 6 tiles loaded together (2 sets)", d, o, side, n);
    run<0, false>("wave per chunk, rolling re-load (4 in flight)", d, o, side, n);
    run<13, false>("wave per chunk, 3 tiles loaded together (2 sets)", d, o, side, n);
    run<1, false>("wave per chunk, 4 tiles loaded together (2 sets)", d, o, side, n);
  }
  return 0;
}
