// scratch/valu_lat.hip -- dependent-chain cost of the few VALU / LDS instructions K7's decode chain is made of (gfx950), with
// 1, 2 and 4 waves per SIMD running the same chain.  cycles = s_memtime ticks per instruction of the chain, per wave.
//   hipcc -O3 --offload-arch=gfx950 -o scratch/valu_lat scratch/valu_lat.hip && scratch/valu_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void k(unsigned long long* out, unsigned seed, int iters) {
  __shared__ unsigned lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (i * 2654435761u) & 0xFFC;  // an address again
  __syncthreads();
  unsigned a = seed + threadIdx.x, b = seed * 3 + 1, s = (threadIdx.x & 7) + 1, thr = 0x80000000u;
  unsigned long long w = ((unsigned long long)a << 32) | b;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) { REP16(asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(w) : "v"(s));) }
    if (OP == 1) { REP16(asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(s));) }
    if (OP == 2) { REP16(asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a) : "v"(s));) }
    if (OP == 3) { REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(s));) }
    if (OP == 4) { REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %2, vcc" : "+v"(a) : "v"(thr), "v"(s) : "vcc");) }
    if (OP == 5) { REP16(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a) : : "memory");) }
    if (OP == 6) { REP16(asm volatile("v_lshlrev_b64 %0, %2, %0\n v_lshrrev_b32 %1, 23, %3\n v_lshl_add_u32 %1, %1, 2, %4\n ds_read_b32 %1, %1\n s_waitcnt lgkmcnt(0)\n v_add_u32_sdwa %2, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"
                                   : "+v"(w), "+v"(a), "+v"(s) : "v"((unsigned)(w >> 32)), "v"(b & 0xFF) : "memory");) }
    if (OP == 7) { REP16(asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a) : "v"(b));) }
    if (OP == 8) { REP16(asm volatile("v_lshrrev_b32 %0, 3, %0\n v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a) : "v"(b));) }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = (t1 - t0) + (a == 0x12345u) + (w == 0x77ull);
}
template <int OP>
void run(const char* name, int per_iter) {
  unsigned long long* d; hipMalloc(&d, 256 * 16 * 8);
  for (int waves : {4, 8, 16}) {  // per CU: 1, 2, 4 per SIMD
    const int iters = 2000;
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(waves * 64), 0, 0, d, 12345u, iters);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(waves * 64), 0, 0, d, 12345u, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * waves);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    printf("%-58s %d waves/SIMD: %6.1f cycles per chain step\n", name, waves / 4, s / h.size() / iters / 16.0 / per_iter * per_iter);
  }
  hipFree(d);
}
int main() {
  run<0>("v_lshlrev_b64 (dependent)", 1);
  run<1>("v_alignbit_b32 (dependent)", 1);
  run<2>("v_lshlrev_b32 (dependent)", 1);
  run<3>("v_add_u32 (dependent)", 1);
  run<7>("v_add_u32_sdwa (dependent)", 1);
  run<4>("v_cmp_lt_u32 + v_addc_co_u32 (dependent pair)", 1);
  run<8>("v_lshrrev_b32 + v_lshl_add_u32 (dependent pair)", 1);
  run<5>("ds_read_b32 + s_waitcnt (dependent, random addresses)", 1);
  run<6>("K7's step: lshl_b64, lshr, lshl_add, ds_read, wait, add_sdwa", 1);
  return 0;
}
