"""K6's class walk (csrc/ghf_decode.hip: k6_jump and the table kernel's chain loop) is plain C++ apart from its qualifiers:
compiled for the HOST here and run against single-step decoding on random masks, for both class lengths.  (The masks
themselves -- k6_classes: bit-matrix transposes with v_perm_b32 / v_alignbit_b32 -- are GPU code; they are pinned by the
-m gpu stream tests, and their arithmetic by scratch-free Python models of the same steps in this file.)"""
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "golden-huffman_amd", "csrc", "ghf_decode.hip")

MAIN = r'''
template <int CLEN> void run(const uint64_t* F, const uint64_t* G, uint32_t* land, uint32_t* cnt, uint32_t* eo) {
  constexpr int NCH = CLEN + 1;
  const uint64_t* Fl = F; const uint64_t* Gl = G;
%(block)s
  for (int e = 0; e < NCH; ++e) { land[e] = p[e] - 512u; cnt[e] = nl[e]; }
  *eo = eofs;
}
template <int CLEN> int check(const uint64_t* F, const uint64_t* G) {
  uint32_t land[9], cnt[9], eo;
  run<CLEN>(F, G, land, cnt, &eo);
  for (int e = 0; e <= CLEN; ++e) {
    uint32_t q = e, n = 0; bool ee = false;
    while (q < 512) { uint32_t c = q & 7, i = q >> 3; if ((F[c] >> (63 - i)) & 1) { if ((G[c] >> (63 - i)) & 1) ee = true; q += CLEN + 1; ++n; } else q += CLEN; }
    if (q - 512 != land[e] || n != cnt[e] || ee != (bool)((eo >> e) & 1)) { printf("MISMATCH L=%%d e=%%d land %%u/%%u n %%u/%%u\n", CLEN, e, land[e], q - 512, cnt[e], n); return 1; }
  }
  return 0;
}
int main() {
  srand(11);
  for (int trial = 0; trial < 60000; ++trial) {
    uint64_t F[8], G[8]; int dens = rand() %% 4;
    for (int r = 0; r < 8; ++r) { uint64_t f = 0, g = 0; for (int i = 0; i < 64; ++i) { int set = dens == 0 ? (rand() %% 16 == 0) : dens == 1 ? (rand() %% 4 == 0) : dens == 2 ? (rand() %% 64 == 0) : (rand() %% 2); if (set) { f |= 1ull << (63 - i); if (rand() & 1) g |= 1ull << (63 - i); } } F[r] = f; G[r] = g; }
    if (check<4>(F, G) || check<8>(F, G)) return 1;
  }
  printf("ok\n");
  return 0;
}
'''


def test_class_walk_matches_single_steps_on_the_host(tmp_path):
    s = open(SRC).read()
    a = s.index("template <int L, bool WITH_G, typename FM, typename GM>")
    b = s.index("__global__ __launch_bounds__(kK6Threads, 4) void k_sync_pass(SyncParams P) {")
    fn = s[a:b].replace("__device__ __forceinline__", "static inline")
    a2 = s.index("        // a chain = the bit p at which its next code begins; >= 512: it has left the subsequence\n        uint32_t p[NCH], nl[NCH], eofs = 0;")
    b2 = s.index("        uint4 row;\n        if (CLEN == 8) {")
    block = s[a2:b2].replace("kK6Threads", "1").replace("Gl[c * 64]", "Gl[c]").replace("* 64]", "]")
    prog = "#include <cstdint>\n#include <cstdio>\n#include <cstdlib>\n" + fn + MAIN % {"block": block}
    src = tmp_path / "walk.cc"
    src.write_text(prog)
    exe = tmp_path / "walk"
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", "-o", str(exe), str(src)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-400:]


def _transpose8(hi, lo):
    M = 0xFFFFFFFF

    def bfi(m, a, b):
        return ((m & a) | (~m & b)) & M

    def dswap(x, m, d):
        return bfi(m, x >> d, bfi((m << d) & M, (x << d) & M, x))

    hi = dswap(dswap(hi, 0x00AA00AA, 7), 0x0000CCCC, 14)
    lo = dswap(dswap(lo, 0x00AA00AA, 7), 0x0000CCCC, 14)
    return bfi(0xF0F0F0F0, hi, lo >> 4), bfi(0xF0F0F0F0, (hi << 4) & M, lo)


def test_bit_matrix_transpose_steps():
    """the delta-swap / v_bfi_b32 sequence of k6_classes transposes 8 bytes x 8 bits (rows = bytes, first byte on top)"""
    rng = np.random.default_rng(1)
    for _ in range(2000):
        b = rng.integers(0, 256, 8)
        hi = int(b[0]) << 24 | int(b[1]) << 16 | int(b[2]) << 8 | int(b[3])
        lo = int(b[4]) << 24 | int(b[5]) << 16 | int(b[6]) << 8 | int(b[7])
        th, tl = _transpose8(hi, lo)
        t = (th << 32) | tl
        for s_ in range(8):  # plane s = bit s (from the top) of every byte, byte 0 first
            want = 0
            for k in range(8):
                want = (want << 1) | ((int(b[k]) >> (7 - s_)) & 1)
            assert (t >> (56 - 8 * s_)) & 0xFF == want
