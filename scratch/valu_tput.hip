// scratch/valu_tput.hip -- ISSUE cost (throughput) of the VALU instructions K5 / K7 are made of, gfx950: every wave runs eight
// independent chains of one instruction, 1 / 2 / 4 waves per SIMD; printed: SIMD cycles per wave-instruction.
//   hipcc -O3 --offload-arch=gfx950 -o scratch/valu_tput scratch/valu_tput.hip && scratch/valu_tput
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define R8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
template <int OP>
__global__ void k(unsigned long long* out, unsigned seed, int iters) {
  unsigned a[8], b = seed * 3 + 1, s = (threadIdx.x & 7) + 1;
  unsigned long long w[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * (i + 1); w[i] = ((unsigned long long)a[i] << 32) | (b + i); }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#define M0(i) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(w[i]) : "v"(s));
#define M1(i) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(s));
#define M2(i) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(a[i]) : "v"(s));
#define M3(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
#define M4(i) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(b));
#define M5(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b));
#define M6(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(s));
#define M7(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
#define M8(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(s));
#define M9(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(s));
#define M10(i) asm volatile("v_bfe_u32 %0, %0, %1, 5" : "+v"(a[i]) : "v"(s));
#define M11(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
#define M12(i) asm volatile("v_lshrrev_b32 %0, 7, %0" : "+v"(a[i]));
#define M13(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
#define M14(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(s) : "vcc");
    if (OP == 0) { R8(M0) R8(M0) } if (OP == 1) { R8(M1) R8(M1) } if (OP == 2) { R8(M2) R8(M2) } if (OP == 3) { R8(M3) R8(M3) }
    if (OP == 4) { R8(M4) R8(M4) } if (OP == 5) { R8(M5) R8(M5) } if (OP == 6) { R8(M6) R8(M6) } if (OP == 7) { R8(M7) R8(M7) }
    if (OP == 8) { R8(M8) R8(M8) } if (OP == 9) { R8(M9) R8(M9) } if (OP == 10) { R8(M10) R8(M10) } if (OP == 11) { R8(M11) R8(M11) }
    if (OP == 12) { R8(M12) R8(M12) } if (OP == 13) { R8(M13) R8(M13) } if (OP == 14) { R8(M14) R8(M14) }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned x = 0; unsigned long long y = 0;
  for (int i = 0; i < 8; ++i) { x ^= a[i]; y ^= w[i]; }
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = (t1 - t0) + (x == 0x12345u) + (y == 0x77ull);
}
template <int OP>
void run(const char* name) {
  unsigned long long* d; (void)hipMalloc(&d, 256 * 16 * 8);
  printf("%-22s", name);
  for (int waves : {4, 8, 16}) {  // per CU: 1, 2, 4 per SIMD
    const int iters = 1000;
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(waves * 64), 0, 0, d, 12345u, iters);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(waves * 64), 0, 0, d, 12345u, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * waves);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    const double per_wave_instr = s / h.size() / iters / 16.0;
    printf("  %d w/SIMD: %5.2f cyc/instr/wave = %5.2f per SIMD", waves / 4, per_wave_instr, per_wave_instr / (waves / 4));
  }
  printf("\n");
  (void)hipFree(d);
}
int main() {
  run<0>("v_lshlrev_b64"); run<1>("v_alignbit_b32"); run<2>("v_lshlrev_b32"); run<12>("v_lshrrev_b32 imm"); run<3>("v_add_u32"); run<4>("v_add_u32_sdwa");
  run<5>("v_lshl_add_u32"); run<6>("v_perm_b32"); run<7>("v_cndmask_b32"); run<8>("v_or3_b32"); run<9>("v_and_or_b32"); run<10>("v_bfe_u32");
  run<11>("v_cmp_lt_u32"); run<13>("v_mov_b32"); run<14>("v_add_co_u32");
  return 0;
}
