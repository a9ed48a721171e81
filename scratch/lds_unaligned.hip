// scratch/lds_unaligned.hip -- does an 8-byte LDS read at ANY byte address return the eight bytes that start there (gfx950,
// the compute queue's alignment mode as the driver sets it)?  And what does it cost when the lanes' addresses are 64.5 bytes
// apart (K7's lanes on uniform bytes) with and without K7's row padding (16 bytes behind every 128)?
//   hipcc -O3 --offload-arch=gfx950 -o scratch/lds_unaligned scratch/lds_unaligned.hip && scratch/lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void check(uint64_t* out, int* bad) {
  __shared__ __attribute__((aligned(16))) uint8_t buf[4096 + 16];
  for (int i = threadIdx.x; i < 4096 + 16; i += blockDim.x) buf[i] = (uint8_t)(i * 7 + 3);
  __syncthreads();
  int nbad = 0;
  for (int a = threadIdx.x; a < 4096; a += blockDim.x) {
    uint64_t v;
    const uint32_t addr = (uint32_t)(uintptr_t)(buf + a);
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    uint64_t want = 0;
    for (int i = 0; i < 8; ++i) want |= (uint64_t)(uint8_t)((a + i) * 7 + 3) << (8 * i);
    if (v != want) { ++nbad; if (a < 16) out[a] = v; }
  }
  if (nbad) atomicAdd(bad, nbad);
}
template <int PAD>
__global__ void cost(uint64_t* out, int iters, int stride_x2) {
  __shared__ __attribute__((aligned(16))) uint8_t buf[64 * 1024];
  for (int i = threadIdx.x; i < 64 * 1024; i += blockDim.x) buf[i] = (uint8_t)i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t la = (uint32_t)wave * 4096u + (uint32_t)(lane * stride_x2) / 2u;
  uint64_t acc = 0;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const uint32_t a = PAD ? la + ((la >> 7) << 4) : la;
    uint64_t v;
    const uint32_t addr = (uint32_t)(uintptr_t)buf + (a & 0xFFFFu) % (64 * 1024 - 8);
    asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    acc += v;
    la += 7;  // six 9-bit codes later
    if ((it & 7) == 7) la -= 56;
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + wave] = (t1 - t0) + (acc == 0x1234567ull);
}
int main() {
  uint64_t* d; int* bad; (void)hipMalloc(&d, 1 << 20); (void)hipMalloc(&bad, 4); (void)hipMemset(bad, 0, 4); (void)hipMemset(d, 0, 1 << 20);
  hipLaunchKernelGGL(check, dim3(1), dim3(256), 0, 0, d, bad);
  int hb = -1; (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  printf("unaligned ds_read_b64: %d of 4096 byte addresses returned something else than the eight bytes that start there\n", hb);
  for (int pad = 0; pad < 2; ++pad)
    for (int sx2 : {129, 128, 66, 16}) {
      const int iters = 4096;
      if (pad) hipLaunchKernelGGL(cost<1>, dim3(256), dim3(1024), 0, 0, d, iters, sx2); else hipLaunchKernelGGL(cost<0>, dim3(256), dim3(1024), 0, 0, d, iters, sx2);
      (void)hipDeviceSynchronize();
      static uint64_t h[4096]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      double s = 0; for (int i = 0; i < 4096; ++i) s += (double)h[i];
      printf("16 waves per CU, lane stride %.1f B, %s: %.1f cycles per dependent ds_read_b64 per wave (%.2f per CU-read)\n", sx2 / 2.0, pad ? "rows padded (16 B per 128)" : "linear", s / 4096 / iters, s / 4096 / iters / 16);
    }
  return 0;
}
