#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
template <int L, bool WITH_G, typename FM, typename GM>
__device__ __forceinline__ uint32_t k6_jump(FM fmask, GM gmask, uint32_t p, bool& found, bool& eof) {
  const uint32_t a = p & 7u, i = p >> 3;
  if (L == 8) {
    const uint64_t m = fmask(a) << i;
    const uint64_t g = WITH_G ? gmask(a) << i : 0ull;
    found = m != 0;
    const uint32_t z = (uint32_t)__builtin_clzll(m | 1ull);
    eof = WITH_G && found && ((g << z) >> 63);
    return found ? z : 64u - i;
  }
  // L = 4: the chain alternates between class a (steps 0, 2, ..) and class a + 4 mod 8 (steps 1, 3, ..: the same byte when
  // a < 4, the next one otherwise)
  const uint32_t b = (a + 4u) & 7u, ib = i + (a >> 2), ibm = ib & 63u;
  const uint64_t ma = fmask(a) << i;
  const uint64_t mb = ib < 64u ? fmask(b) << ibm : 0ull;
  const uint64_t ga = WITH_G ? gmask(a) << i : 0ull;
  const uint64_t gb = WITH_G ? gmask(b) << ibm : 0ull;
  const uint32_t ta = (uint32_t)__builtin_clzll(ma | 1ull), tb = (uint32_t)__builtin_clzll(mb | 1ull);
  const uint32_t sa = ma ? 2u * ta : 1000u;
  const uint32_t sb = mb ? 2u * tb + 1u : 1000u;
  const bool first_a = sa < sb;
  const uint32_t s = first_a ? sa : sb;
  found = s < 1000u;
  eof = WITH_G && found && (((first_a ? ga << ta : gb << tb) >> 63) != 0);
  return found ? s : (515u - p) >> 2;
}


__global__ void k(const uint64_t* F, const uint64_t* G, const uint32_t* P, uint32_t* out, int n) {
  __shared__ uint64_t Fl[8 * 256];
  __shared__ uint64_t Gl[8 * 256];
  const int t = blockIdx.x * 256 + threadIdx.x;
  for (int r = 0; r < 8; ++r) { Fl[r * 256 + threadIdx.x] = F[(size_t)t * 8 + r]; Gl[r * 256 + threadIdx.x] = G[(size_t)t * 8 + r]; }
  __syncthreads();
  bool found, eof;
  const uint64_t* fl = Fl + threadIdx.x; const uint64_t* gl = Gl + threadIdx.x;
  uint32_t s = k6_jump<4, true>([&](uint32_t c) { return fl[c * 256]; }, [&](uint32_t c) { return gl[c * 256]; }, P[t], found, eof);
  bool f2, e2;
  uint32_t s2 = k6_jump<4, false>([&](uint32_t c) { return fl[c * 256]; }, [](uint32_t) { return 0ull; }, P[t], f2, e2);
  out[t] = s | (found ? 1u << 16 : 0) | (eof ? 1u << 17 : 0) | (s2 != s || f2 != found ? 1u << 20 : 0);
}
int main() {
  const int n = 256 * 256;
  uint64_t *F = (uint64_t*)malloc(n * 64), *G = (uint64_t*)malloc(n * 64); uint32_t* P = (uint32_t*)malloc(n * 4); uint32_t* O = (uint32_t*)malloc(n * 4);
  srand(7);
  for (int t = 0; t < n; ++t) { int dens = rand() % 3; for (int r = 0; r < 8; ++r) { uint64_t f = 0, g = 0; for (int i = 0; i < 64; ++i) { int set = dens == 0 ? (rand() % 16 == 0) : dens == 1 ? (rand() % 4 == 0) : (rand() % 64 == 0); if (set) { f |= 1ull << (63 - i); if (rand() & 1) g |= 1ull << (63 - i); } } F[t * 8 + r] = f; G[t * 8 + r] = g; } P[t] = rand() % 512; }
  uint64_t *dF, *dG; uint32_t *dP, *dO;
  hipMalloc(&dF, n * 64); hipMalloc(&dG, n * 64); hipMalloc(&dP, n * 4); hipMalloc(&dO, n * 4);
  hipMemcpy(dF, F, n * 64, hipMemcpyHostToDevice); hipMemcpy(dG, G, n * 64, hipMemcpyHostToDevice); hipMemcpy(dP, P, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dF, dG, dP, dO, n);
  hipMemcpy(O, dO, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < n; ++t) {
    uint32_t q = P[t], ns = 0; bool nf = false, ne = false;
    while (q < 512) { uint32_t c = q & 7, i = q >> 3; if ((F[t * 8 + c] >> (63 - i)) & 1) { nf = true; ne = (G[t * 8 + c] >> (63 - i)) & 1; break; } q += 4; ++ns; }
    uint32_t exp = ns | (nf ? 1u << 16 : 0) | ((nf && ne) ? 1u << 17 : 0);
    if (O[t] != exp) { if (bad < 10) printf("t=%d p=%u got %x exp %x\n", t, P[t], O[t], exp); ++bad; }
  }
  printf("bad %d of %d\n", bad, n);
}
