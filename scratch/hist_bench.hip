// scratch microbenchmark: what bounds the byte histogram on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE, int THREADS, int REPSHIFT>
__global__ __launch_bounds__(THREADS) void k(const uint4* __restrict__ in, uint64_t nvec, unsigned long long* out) {
  constexpr int REP = 1 << REPSHIFT;
  __shared__ uint32_t lh[256 * REP];
  const uint32_t tid = threadIdx.x;
  const uint32_t rep = tid & (REP - 1);
  for (uint32_t i = tid; i < 256 * REP; i += THREADS) lh[i] = 0;
  __syncthreads();
  uint32_t x = 0;
  const uint64_t stride = (uint64_t)gridDim.x * THREADS;
  uint64_t i = (uint64_t)blockIdx.x * THREADS + tid;
  auto proc = [&](const uint4& v) {
    if (MODE == 0) { x ^= v.x ^ v.y ^ v.z ^ v.w; return; }
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const uint32_t byte = (w[q] >> (8 * b)) & 0xFF;
        if (MODE == 3) x += byte << REPSHIFT | rep;
        else atomicAdd(&lh[(byte << REPSHIFT) | rep], 1u);
      }
    }
  };
  if (i + 3 * stride < nvec) {
    uint4 A = in[i], B = in[i + stride];
    for (; i + 7 * stride < nvec; i += 4 * stride) {
      const uint4 C = in[i + 2 * stride], D = in[i + 3 * stride];
      proc(A); proc(B);
      A = in[i + 4 * stride]; B = in[i + 5 * stride];
      proc(C); proc(D);
    }
    const uint4 C = in[i + 2 * stride], D = in[i + 3 * stride];
    proc(A); proc(B); proc(C); proc(D);
    i += 4 * stride;
  }
  for (; i < nvec; i += stride) proc(in[i]);
  __syncthreads();
  unsigned long long s = x;
  if (MODE == 1 || MODE == 2) for (int j = 0; j < REP; ++j) s += lh[(tid % 256) * REP + j];
  if (s == 0xdeadbeefcafeULL) out[0] = s;
  if (MODE != 0 && MODE != 3 && tid < 256) atomicAdd(&out[tid], s);
}

template <int MODE, int THREADS, int REPSHIFT>
int run(const char* name, const uint4* d, uint64_t nvec, unsigned long long* dout, int grid) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<MODE, THREADS, REPSHIFT>), dim3(grid), dim3(THREADS), 0, 0, d, nvec, dout);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int R = 10;
  for (int r = 0; r < R; ++r) hipLaunchKernelGGL((k<MODE, THREADS, REPSHIFT>), dim3(grid), dim3(THREADS), 0, 0, d, nvec, dout);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= R;
  printf("%-44s grid %5d  %.3f ms  %.1f GB/s\n", name, grid, ms, nvec * 16.0 / ms / 1e6);
  return 0;
}

int main(int argc, char** argv) {
  const uint64_t n = 1ull << 28;
  int sym16 = argc > 1;
  std::vector<uint8_t> h(n);
  uint64_t z = 88172645463325252ull;
  for (uint64_t i = 0; i < n; i += 8) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; uint64_t v = sym16 ? (z & 0x0F0F0F0F0F0F0F0Full) : z; memcpy(&h[i], &v, 8); }
  uint4* d; unsigned long long* dout;
  CK(hipMalloc(&d, n)); CK(hipMalloc(&dout, 4096)); CK(hipMemset(dout, 0, 4096));
  CK(hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice));
  const uint64_t nvec = n / 16;
  printf("data: %s\n", sym16 ? "16 symbols" : "uniform");
  for (int g : {1024, 1280, 2048}) {
    run<0, 256, 5>("loads only, 256 thr", d, nvec, dout, g);
    run<3, 256, 5>("loads + byte extract (VALU only)", d, nvec, dout, g);
    run<1, 256, 5>("ds_add [256][32], 256 thr", d, nvec, dout, g);
  }
  run<1, 256, 3>("ds_add [256][8], 256 thr", d, nvec, dout, 2048);
  run<1, 256, 0>("ds_add [256][1], 256 thr", d, nvec, dout, 2048);
  run<1, 512, 5>("ds_add [256][32], 512 thr", d, nvec, dout, 1024);
  run<1, 1024, 5>("ds_add [256][32], 1024 thr", d, nvec, dout, 512);
  run<1, 1024, 4>("ds_add [256][16], 1024 thr", d, nvec, dout, 512);
  run<0, 1024, 5>("loads only, 1024 thr", d, nvec, dout, 512);
  return 0;
}
