// scratch/ifetch2.hip -- where is the cliff?  Loop bodies of N x 64 VALU instructions (8-byte v_lshl_add_u32 / 4-byte v_add_u32),
// 16 and 24 waves per CU on every CU.  Printed: SIMD cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define I4 asm volatile("v_add_u32 %0, %0, %4\nv_add_u32 %1, %1, %4\nv_add_u32 %2, %2, %4\nv_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));
#define J4 asm volatile("v_lshl_add_u32 %0, %0, 1, %4\nv_lshl_add_u32 %1, %1, 1, %4\nv_lshl_add_u32 %2, %2, 1, %4\nv_lshl_add_u32 %3, %3, 1, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define B64I R16(I4)
#define B64J R16(J4)
template <int N64, bool WIDE>
__global__ void k(unsigned long long* out, unsigned seed, int trips) {
  unsigned a = seed + threadIdx.x, b = a * 3, c = a * 5, d = a * 7, s = (threadIdx.x & 7) + 1;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < trips; ++it) {
#pragma unroll
    for (int r = 0; r < N64; ++r) { if (WIDE) { B64J } else { B64I } }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = (t1 - t0) + ((a ^ b ^ c ^ d) == 0x12345u);
}
template <int N64, bool WIDE>
void run() {
  unsigned long long* dbuf; (void)hipMalloc(&dbuf, 256 * 32 * 8);
  printf("%4d x %d B = %5.1f KB loop:", N64 * 64, WIDE ? 8 : 4, N64 * 64 * (WIDE ? 8 : 4) / 1024.0);
  for (int waves : {16, 24}) {
    const int trips = 4096 / N64;
    hipLaunchKernelGGL((k<N64, WIDE>), dim3(256), dim3(waves * 64 > 1024 ? 512 : waves * 64), 0, 0, dbuf, 12345u, trips);
    // (24 waves per CU = three workgroups of 512 threads: launch 768 of them)
    hipLaunchKernelGGL((k<N64, WIDE>), dim3(waves == 24 ? 768 : 256), dim3(waves == 24 ? 512 : waves * 64), 0, 0, dbuf, 12345u, trips);
    (void)hipDeviceSynchronize();
    const int nw = waves == 24 ? 768 * 8 : 256 * waves;
    std::vector<unsigned long long> h(nw);
    (void)hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += (double)v;
    const double per = sum / h.size() / trips / (N64 * 64.0);
    printf("   %2d waves/CU: %5.2f cyc/instr/wave = %5.2f per SIMD", waves, per, per / (waves / 4));
  }
  printf("\n");
  (void)hipFree(dbuf);
}
int main() {
  run<4, true>(); run<12, true>(); run<16, true>(); run<17, true>(); run<18, true>(); run<20, true>(); run<24, true>(); run<28, true>(); run<32, true>();
  run<16, false>(); run<18, false>(); run<20, false>(); run<24, false>(); run<32, false>();
  return 0;
}
