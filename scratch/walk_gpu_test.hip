#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
constexpr int kK6Threads = 1024;
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


struct Lds { uint64_t f[8 * 1024]; uint64_t g[16 * 8 * 64 + 16 * 154]; };
template <int CLEN>
__global__ __launch_bounds__(1024, 4) void k(const uint64_t* Fg, const uint64_t* Gg, uint32_t* out) {
  __shared__ Lds S;
  constexpr int NCH = CLEN + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t t = (size_t)blockIdx.x * 1024 + tid;
  uint64_t F[8], G[8];
  for (int r = 0; r < 8; ++r) { F[r] = Fg[t * 8 + r]; G[r] = Gg[t * 8 + r]; }
  uint64_t* const Fl = S.f + tid;
  uint64_t* const Gl = S.g + wave * (5328 / 8) + lane;
  for (int r = 0; r < 8; ++r) { Fl[r * kK6Threads] = F[r]; Gl[r * 64] = G[r]; }
        // a chain = the bit p at which its next code begins; >= 512: it has left the subsequence
        uint32_t p[NCH], nl[NCH], eofs = 0;
        bool any = false;
#pragma unroll
        for (int e = 0; e < NCH; ++e) {  // first round: every chain stands at its entry offset, the masks still in registers
          bool found, eof;
          const uint32_t sh = k6_jump<CLEN, true>([&](uint32_t c) { return F[c]; }, [&](uint32_t c) { return G[c]; }, (uint32_t)e, found, eof);
          const uint32_t q = (uint32_t)e + (uint32_t)CLEN * sh;  // the long code (found), or the first position behind the subsequence
          p[e] = found ? q + (uint32_t)CLEN + 1u : q;
          nl[e] = found ? 1u : 0u;
          eofs |= eof ? (1u << e) : 0u;
          any |= p[e] < 512u;
        }
        while (any) {
          any = false;
#pragma unroll
          for (int e = 0; e < NCH; ++e) {
            const bool act = p[e] < 512u;
            const uint32_t pe = act ? p[e] : 0u;
            bool found, eof;
            const uint32_t sh = k6_jump<CLEN, true>([&](uint32_t c) { return Fl[c * kK6Threads]; }, [&](uint32_t c) { return Gl[c * 64]; }, pe, found, eof);
            const uint32_t q = pe + (uint32_t)CLEN * sh;
            p[e] = act ? (found ? q + (uint32_t)CLEN + 1u : q) : p[e];
            nl[e] += (act && found) ? 1u : 0u;
            eofs |= (act && eof) ? (1u << e) : 0u;
            any |= p[e] < 512u;
          }
        }

  for (int e = 0; e < NCH; ++e) { out[t * 16 + e] = p[e]; out[t * 16 + 5 + e] = nl[e]; }
  out[t * 16 + 15] = eofs;
}
int main() {
  const int n = 256 * 1024;
  uint64_t *F = (uint64_t*)malloc((size_t)n * 64), *G = (uint64_t*)malloc((size_t)n * 64); uint32_t* O = (uint32_t*)malloc((size_t)n * 64);
  srand(13);
  for (int t = 0; t < n; ++t) { int dens = rand() % 3; for (int r = 0; r < 8; ++r) { uint64_t f = 0, g = 0; for (int i = 0; i < 64; ++i) { int set = dens == 0 ? (rand() % 16 == 0) : dens == 1 ? (rand() % 4 == 0) : (rand() % 64 == 0); if (set) { f |= 1ull << (63 - i); if (rand() & 1) g |= 1ull << (63 - i); } } F[t * 8 + r] = f; G[t * 8 + r] = g; } }
  uint64_t *dF, *dG; uint32_t* dO;
  (void)hipMalloc(&dF, (size_t)n * 64); (void)hipMalloc(&dG, (size_t)n * 64); (void)hipMalloc(&dO, (size_t)n * 64);
  (void)hipMemcpy(dF, F, (size_t)n * 64, hipMemcpyHostToDevice); (void)hipMemcpy(dG, G, (size_t)n * 64, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k<4>, dim3(n / 1024), dim3(1024), 0, 0, dF, dG, dO);
  (void)hipMemcpy(O, dO, (size_t)n * 64, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < n; ++t) for (int e = 0; e < 5; ++e) {
    uint32_t q = e, nn = 0; bool ee = false;
    while (q < 512) { uint32_t c = q & 7, i = q >> 3; if ((F[t * 8 + c] >> (63 - i)) & 1) { if ((G[t * 8 + c] >> (63 - i)) & 1) ee = true; q += 5; ++nn; } else q += 4; }
    const bool ge = (O[t * 16 + 15] >> e) & 1;
    if (O[t * 16 + e] != q || O[t * 16 + 5 + e] != nn || ge != ee) { if (bad < 10) printf("t=%d e=%d p %u/%u n %u/%u eof %d/%d\n", t, e, O[t * 16 + e], q, O[t * 16 + 5 + e], nn, ge, ee); ++bad; }
  }
  printf("bad %d of %d\n", bad, n * 5);
}
