// scratch/hist_wave.hip -- prototype for round 4: K1 with WAVE-private histograms (no workgroup barrier per chunk).
// One wave = one chunk (as K5), 256 bins x R replicas per wave in LDS, four 16-byte loads in flight per lane, the chunk's
// 256 counts written at the end + added into 32 global replicas.  Timed against the chunk structure of the shipped K1 on
// uniform bytes, 16 values and a constant.   hipcc --offload-arch=gfx950 -O3 -o scratch/hist_wave scratch/hist_wave.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4_stream __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_stream(const void* p) {
  const u32x4_stream x = __builtin_nontemporal_load(reinterpret_cast<const u32x4_stream*>(p));
  return make_uint4(x.x, x.y, x.z, x.w);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int R>
__global__ __launch_bounds__(512) void k_hist_wave(const uint8_t* __restrict__ in, uint64_t n, uint32_t chunk, uint32_t nchunks,
                                                   uint32_t* __restrict__ chunk_hist, unsigned long long* __restrict__ acc) {
  __shared__ uint32_t lh[8][256 * R];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* h = lh[wave];
  const uint32_t rep = (uint32_t)lane & (R - 1);
  for (uint32_t c = blockIdx.x * 8 + wave; c < nchunks; c += gridDim.x * 8) {
    for (int i = lane; i < 256 * R; i += 64) h[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint4* p = reinterpret_cast<const uint4*>(in + (uint64_t)c * chunk) + lane;
    const uint32_t V = chunk >> 10;  // 1 KiB rows per wave
    auto add = [&](const uint4& v) {
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
      if (v.x == v.y && v.x == v.z && v.x == v.w && ((v.x >> 8) | (v.x << 24)) == v.x) {  // sixteen equal bytes: one add of 16
        atomicAdd(&h[((v.x & 0xFFu) * R) | rep], 16u);
        return;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        atomicAdd(&h[((w[k] & 0xFFu) * R) | rep], 1u);
        atomicAdd(&h[(((w[k] >> 8) & 0xFFu) * R) | rep], 1u);
        atomicAdd(&h[(((w[k] >> 16) & 0xFFu) * R) | rep], 1u);
        atomicAdd(&h[((w[k] >> 24) * R) | rep], 1u);
      }
    };
    uint4 A = load_stream(p), B = load_stream(p + 64);
    uint32_t j = 0;
    for (; j + 4 <= V; j += 4) {
      const uint4 C = load_stream(p + 64 * (j + 2)), D = load_stream(p + 64 * (j + 3));
      add(A); add(B);
      if (j + 4 < V) { A = load_stream(p + 64 * (j + 4)); B = load_stream(p + 64 * (j + 5 < V ? j + 5 : j + 4)); }
      add(C); add(D);
    }
    for (; j < V; ++j) add(load_stream(p + 64 * j));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // bins 4 lane .. 4 lane + 3: sum the replicas
    uint32_t s[4] = {0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < 4; ++b)
      for (int r = 0; r < R; ++r) s[b] += h[(4 * lane + b) * R + ((r + lane) & (R - 1))];
    reinterpret_cast<uint4*>(chunk_hist + (uint64_t)c * 256)[lane] = make_uint4(s[0], s[1], s[2], s[3]);
    unsigned long long* a = acc + (uint64_t)(c & 31u) * 256 + 4 * lane;
#pragma unroll
    for (int b = 0; b < 4; ++b) if (s[b]) atomicAdd(&a[b], (unsigned long long)s[b]);
  }
}

__global__ void k_fill(uint8_t* p, uint64_t n, int kind) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    p[i] = kind == 0 ? (uint8_t)x : kind == 1 ? (uint8_t)(x & 15) : kind == 2 ? 0 : (uint8_t)((x & 0xFF) < 200 ? 0 : (x >> 8));
  }
}
template <int R> float run(const uint8_t* d, uint64_t n, uint32_t chunk, uint32_t* ch, unsigned long long* acc, std::vector<unsigned long long>& out) {
  const uint32_t nchunks = (uint32_t)(n / chunk);
  CK(hipMemset(acc, 0, 32 * 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 20;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_hist_wave<R>, dim3(768), dim3(512), 0, 0, d, n, chunk, nchunks, ch, acc);
  CK(hipMemset(acc, 0, 32 * 256 * 8));
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_hist_wave<R>, dim3(768), dim3(512), 0, 0, d, n, chunk, nchunks, ch, acc);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(32 * 256); CK(hipMemcpy(h.data(), acc, 32 * 256 * 8, hipMemcpyDeviceToHost));
  out.assign(256, 0);
  for (int r = 0; r < 32; ++r) for (int b = 0; b < 256; ++b) out[b] += h[r * 256 + b] / reps;
  return ms / reps;
}
int main() {
  for (uint64_t mib : {256ull, 4096ull}) {
    const uint64_t n = mib << 20;
    const uint32_t chunk = mib == 256 ? 45056 : 700416;
    uint8_t* d; uint32_t* ch; unsigned long long* acc;
    CK(hipMalloc(&d, n)); CK(hipMalloc(&ch, (n / chunk + 1) * 1024)); CK(hipMalloc(&acc, 32 * 256 * 8));
    const char* names[4] = {"uniform", "16 values", "constant", "78% zeros"};
    for (int kind = 0; kind < 4; ++kind) {
      hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, d, n, kind);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> h4, h8;
      const float t8 = run<8>(d, n, chunk, ch, acc, h8);
      const float t4 = run<4>(d, n, chunk, ch, acc, h4);
      unsigned long long tot = 0; for (auto x : h8) tot += x;
      printf("%4llu MiB %-10s  R=8 %.4f ms  R=4 %.4f ms   (sum %llu of %llu, bin0 %llu)\n", (unsigned long long)mib, names[kind], t8, t4, tot,
             (unsigned long long)(n / chunk) * chunk, h8[0]);
    }
    CK(hipFree(d)); CK(hipFree(ch)); CK(hipFree(acc));
  }
}
