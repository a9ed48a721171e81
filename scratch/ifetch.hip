// scratch/ifetch.hip -- is a long straight-line instruction stream (K5's / K7's unrolled passes: 6..8 KB per pass) slower to
// issue than the same instructions in a short loop?  16 waves per CU on every CU run N VALU instructions either as a 16-
// instruction loop body or as ONE block of 2048 (8 KB / 16 KB of code), four independent chains per wave.
//   hipcc -O3 --offload-arch=gfx950 -o scratch/ifetch scratch/ifetch.hip && scratch/ifetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define I4(op) asm volatile(op " %0, %0, %4\n" op " %1, %1, %4\n" op " %2, %2, %4\n" op " %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));
#define J4(op) asm volatile(op " %0, %0, 1, %4\n" op " %1, %1, 1, %4\n" op " %2, %2, 1, %4\n" op " %3, %3, 1, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s));
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R64(x) R4(R16(x))

#define R512(x) R4(R4(R4(R4(R4(x))))) 
template <int MODE>
__global__ void k(unsigned long long* out, unsigned seed, int n16) {
  unsigned a = seed + threadIdx.x, b = a * 3, c = a * 5, d = a * 7, s = (threadIdx.x & 7) + 1;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0) { for (int it = 0; it < n16; ++it) { R4(I4("v_add_u32")) } }                      // 16 instructions (64 B) per trip
  if (MODE == 1) { for (int it = 0; it < n16 / 128; ++it) { R512(I4("v_add_u32")) } }              // 2048 instructions (8 KB) per trip
  if (MODE == 2) { for (int it = 0; it < n16; ++it) { R4(J4("v_lshl_add_u32")) } }
  if (MODE == 3) { for (int it = 0; it < n16 / 128; ++it) { R512(J4("v_lshl_add_u32")) } }
  if (MODE == 5) { for (int it = 0; it < n16 / 4; ++it) { R16(J4("v_lshl_add_u32")) } }            // 64 x 8 B = 512 B
  if (MODE == 6) { for (int it = 0; it < n16 / 8; ++it) { R16(J4("v_lshl_add_u32")) R16(J4("v_lshl_add_u32")) } }  // 128 x 8 = 1 KB
  if (MODE == 7) { for (int it = 0; it < n16 / 16; ++it) { R64(J4("v_lshl_add_u32")) } }           // 256 x 8 = 2 KB
  if (MODE == 8) { for (int it = 0; it < n16 / 32; ++it) { R64(J4("v_lshl_add_u32")) R64(J4("v_lshl_add_u32")) } }  // 512 x 8 = 4 KB
  if (MODE == 9) { for (int it = 0; it < n16 / 2; ++it) { R4(J4("v_lshl_add_u32")) R4(J4("v_lshl_add_u32")) } }     // 32 x 8 B = 256 B (same as 4, check)
  if (MODE == 10) { for (int it = 0; it < n16 / 8; ++it) { R16(I4("v_add_u32")) R16(I4("v_add_u32")) } }            // 128 x 4 B = 512 B
  if (MODE == 12) { for (int it = 0; it < n16 / 64; ++it) { R64(I4("v_add_u32")) R64(I4("v_add_u32")) R64(I4("v_add_u32")) R64(I4("v_add_u32")) } }  // 1024 x 4 B = 4 KB
  if (MODE == 13) { for (int it = 0; it < n16 / 64; ++it) { R64(J4("v_lshl_add_u32")) R64(J4("v_lshl_add_u32")) R64(J4("v_lshl_add_u32")) R64(J4("v_lshl_add_u32")) } }  // 1024 x 8 B = 8 KB
  if (MODE == 14) { for (int it = 0; it < n16 / 96; ++it) { R512(I4("v_add_u32")) } }  // placeholder: 2048 again, trip count differs
  if (MODE == 15) { for (int it = 0; it < n16 / 48; ++it) { R64(J4("v_lshl_add_u32")) R64(J4("v_lshl_add_u32")) R64(J4("v_lshl_add_u32")) } }  // 768 x 8 B = 6 KB
  if (MODE == 11) { for (int it = 0; it < n16 / 32; ++it) { R64(I4("v_add_u32")) R64(I4("v_add_u32")) } }           // 512 x 4 B = 2 KB
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = (t1 - t0) + ((a ^ b ^ c ^ d) == 0x12345u);
}
template <int MODE>
void run(const char* name) {
  unsigned long long* dbuf; (void)hipMalloc(&dbuf, 256 * 16 * 8);
  printf("%-44s", name);
  for (int waves : {4, 8, 16}) {
    const int n16 = 128 * 40;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(waves * 64), 0, 0, dbuf, 12345u, n16);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(waves * 64), 0, 0, dbuf, 12345u, n16);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * waves);
    (void)hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    const double per = s / h.size() / n16 / 16.0;
    printf("  %d w/SIMD: %5.2f cyc/instr/wave = %5.2f per SIMD", waves / 4, per, per / (waves / 4));
  }
  printf("\n");
  (void)hipFree(dbuf);
}
int main() {
  run<0>("v_add_u32 x16 = 64 B loop");
  run<10>("v_add_u32 x128 = 512 B loop");
  run<11>("v_add_u32 x512 = 2 KB loop");
  run<12>("v_add_u32 x1024 = 4 KB loop");
  run<1>("v_add_u32 x2048 = 8 KB loop");
  run<2>("v_lshl_add_u32 (8 B) x16 = 128 B loop");
  run<9>("v_lshl_add_u32 x32 = 256 B loop");
  run<5>("v_lshl_add_u32 x64 = 512 B loop");
  run<6>("v_lshl_add_u32 x128 = 1 KB loop");
  run<7>("v_lshl_add_u32 x256 = 2 KB loop");
  run<8>("v_lshl_add_u32 x512 = 4 KB loop");
  run<15>("v_lshl_add_u32 x768 = 6 KB loop");
  run<13>("v_lshl_add_u32 x1024 = 8 KB loop");
  run<3>("v_lshl_add_u32 x2048 = 16 KB loop");
  return 0;
}
