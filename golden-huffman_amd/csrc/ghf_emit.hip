// golden-huffman_amd/csrc/ghf_emit.hip -- K5, the second pass of the two-pass bit packer (gfx950, wave64).
//
// Replaces CanonicalHuffEncoder::encode_file / encode_each_byte and Buffer::write_bits / write_bit / flush_bits
// (reference include/canonical_huff_encoder.cc:245-285, utils/include/buffer.h:241-248,277-280,290-295) and, when
// asked to, write_encode_info (canonical_huff_encoder.cc:210-242).  ONE launch, no pre-zeroed output, no global
// atomics:
//
//   * one wave owns one chunk (K4 gave every chunk its start bit) and streams through it 1 KiB at a time; lane l
//     loads ITS 16 contiguous symbols with one coalesced 16-byte load and looks the 16 codes up in a 32x replicated
//     LDS table (replica = lane % 32: conflict-free for any data);
//   * four neighbouring symbols are fused into one item of at most 64 bits (codes <= 16 bits; two symbols for longer
//     codes) and a DPP prefix sum of the items' lengths gives every lane its bit position in the wave's LDS bit string;
//   * codes <= 16 bits (every BASELINE config): the lane's four items are concatenated IN REGISTERS into its whole run
//     of <= 256 bits, shifted to the bit phase of its position, and stored with PLAIN ds_write_b32 -- every dword of the
//     bit string is written exactly once by the first lane that touches it; only a lane's FIRST dword, which it may share
//     with the lanes in front of it, is OR-ed in (one ds_or_b32 per lane and tile instead of twelve, nothing to clear
//     afterwards: round 2's PMC had 49 % of this kernel's LDS cycles as bank-conflict cycles of those atomics).
//     Longer codes (rare) keep the three ds_or_b32 per item;
//   * no per-lane serial accumulator, no data-dependent branch: the hot loop is straight-line code;
//   * completed 16-byte units are copied out coalesced; the incomplete last unit is carried to the next tile;
//   * the 16-byte unit a chunk shares with its predecessor belongs to the LATER chunk: its wave re-encodes the last
//     symbols of the previous chunk (or takes the header bytes) to fill the bits in front of its own first code, so
//     every output byte is written exactly once, by one plain store.
#include "ghf_device.h"

namespace ghf {

constexpr int kEmitTabWords = 256 * 32;  // 32 KiB: [256][32] x u32 (len << 16 | code) or [256][16] x u64 (len << 32 | code)

struct WaveOut {
  uint32_t* flat;      // the workgroup's whole staging array (every wave ORs through the same base: one address form)
  uint32_t* st;        // this wave's part of it
  uint32_t bit0;       // bit index of st[0] inside `flat`
  uint4* out_units;    // output as 16-byte units
  uint64_t unit_base;  // unit index (in out) of staging unit 0
  uint32_t carry;      // valid bits at the front of the staging area (< 128)
};

// OR the `len` low bits of q (len <= 64, q < 2^len) into the bit string at bit position p (MSB first inside every word)
__device__ __forceinline__ void deposit64(uint32_t* flat, uint32_t p, uint64_t q, uint32_t len) {
  const uint64_t V = q << ((64u - len) & 63u);  // left-justified (len == 0: q == 0)
  const uint32_t hi = (uint32_t)(V >> 32), lo = (uint32_t)V;
  const uint32_t s = p & 31u;
  uint32_t* w = flat + (p >> 5);
  atomicOr(w, hi >> s);
  atomicOr(w + 1, alignbit(hi, lo, s));
  atomicOr(w + 2, alignbit(lo, 0u, s));
}

// One tile: NI items per lane (in stream order) -> bits in the staging area -> whole units to HBM.
// DRAIN (main loop only): right before the first global store of the tile, wait until at most ONE vector-memory
// operation is outstanding.  In program order the outstanding ones are: the load of tile t+1 (issued two tiles ago),
// the stores of tile t-1, the load of tile t+2 (issued when this tile started) -- so this guarantees tile t+1's data
// and retires the old stores while tile t+2 stays in flight, and the compiler derives no wait of its own from a
// store count it would have to guess.
template <int NI, bool DRAIN>
__device__ __forceinline__ uint32_t emit_items(WaveOut& W, const uint64_t (&q)[NI], const uint32_t (&l)[NI], bool mine, int lane,
                                               uint32_t* seg_dst, uint32_t seg_base, uint64_t* blk_dst, uint64_t blk_val) {
  uint32_t T = 0;  // mine == false: this lane sits the pass out (its items count as empty)
#pragma unroll
  for (int i = 0; i < NI; ++i) T += l[i];
  if (!mine) T = 0;
  const uint32_t incl = wave_incl_scan_u32(T);
  const uint32_t excl = incl - T;
  const uint32_t total = wave_last_u32(incl);
  uint32_t p = W.bit0 + W.carry + excl;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    deposit64(W.flat, p, mine ? q[i] : 0ull, mine ? l[i] : 0u);
    p += l[i];
  }
  wave_sync();
  const uint32_t endbits = W.carry + total;
  const uint32_t U = endbits >> 7;  // <= 128
  if (DRAIN) __builtin_amdgcn_s_waitcnt(0x0F71);  // vmcnt(1)
  if (seg_dst) *seg_dst = seg_base + incl;  // side-car: where this lane's segment ends, relative to its block
  if (blk_dst) *blk_dst = blk_val;          // side-car: where the block begins (one lane, every fourth tile)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t j = (uint32_t)lane + 64u * h;
    if (j < U) {
      uint4* su = reinterpret_cast<uint4*>(W.st) + j;
      uint4 v = *su;
      *su = make_uint4(0, 0, 0, 0);
      v.x = bswap32(v.x); v.y = bswap32(v.y); v.z = bswap32(v.z); v.w = bswap32(v.w);
      W.out_units[W.unit_base + j] = v;  // (a non-temporal hint here: no gain up to 1 GiB, 7 % slower at 4 GiB -- K7 reads this soon)
    }
  }
  wave_sync();
  if (U) {  // the incomplete unit moves to the front
    if (lane < 4) {
      const uint32_t t = W.st[4 * U + lane];
      W.st[4 * U + lane] = 0;
      W.st[lane] = t;
    }
    W.unit_base += U;
  }
  W.carry = endbits & 127u;
  wave_sync();
  return total;
}

// ---- codes <= 16 bits: the lane's whole run leaves as plain stores -------------------------------------------------
// ML = the longest code the instantiation handles (9, 12 or 16).  A lane's four items (right-justified, <= 4 * ML bits
// each) become TWO left-justified strings of <= 8 * ML bits, its HALVES (eight symbols each):
//   V_k = item k left-justified in 64 bits;  P0 = V0 ++ V1,  P1 = V2 ++ V3  (two 64-bit shifts each);
//   rr[h][] = P_h >> (p_h & 31): the NP + 1 dwords of the staging area the half touches, from dword p_h >> 5 on
// (concatenating the halves in registers as well costs a select network -- registers cannot be indexed -- of 40 VALU
// instructions per lane and tile for two LDS stores less: measured, VALU is what this kernel is short of).
// Ownership: dword d of the bit string belongs to the FIRST half that touches it.  That half stores it whole (its own bits,
// zeros behind them); every later half that begins inside d ORs its first dword in afterwards (LDS executes a wave's
// instructions in order).  So: rr[1..nw-1] are plain stores, rr[0] is a plain store when the half begins on a dword
// boundary and a ds_or_b32 otherwise.  The tile's very first dword holds the carry of the previous tile (or the bits
// in front of the chunk), stored the same way.  Nothing is cleared: a dword is complete before it is copied out, and
// what lies behind the last half is overwritten by its first toucher in the next tile.  Halves with no bits (ragged last
// tile, the end-mark pass) store nothing; any number of halves may share a dword.
template <int ML>
struct WinGeom {
  static constexpr int NP = (8 * ML + 31) / 32;   // dwords of a half (two items): 3, 3, 4
};

// X (lx valid bits, left-justified) followed by Y (left-justified): the first 128 bits of the concatenation.
// lx == 0 implies Y == 0 here (the valid symbols of a lane are a prefix of its sixteen).
template <int ML>
__device__ __forceinline__ void cat_left64(uint64_t X, uint32_t lx, uint64_t Y, uint32_t (&P)[WinGeom<ML>::NP]) {
  uint64_t ys = Y >> (lx & 63u);
  if (ML > 15) ys = (lx == 64u) ? 0ull : ys;  // (a shift by 64 is a shift by 0 to the hardware)
  const uint64_t hi = X | ys;
  const uint64_t lo = Y << ((64u - lx) & 63u);
  P[0] = (uint32_t)(hi >> 32);
  P[1] = (uint32_t)hi;
  P[2] = (uint32_t)(lo >> 32);
  if (WinGeom<ML>::NP > 3) P[WinGeom<ML>::NP - 1] = (uint32_t)lo;
}

// KIND 0: any tile (ragged, the end-mark pass; lanes may be empty), every store under its own condition.
// KIND 1 / 2: a full tile of the main loop, with / without side-car.  All global stores are then UNCONDITIONAL straight-line
// instructions (a lane with nothing to store repeats its neighbour's store: same address, same value), so that the
// compiler can count them: gfx9 has ONE in-order counter for loads and stores, and the wait for a tile that was
// requested four tiles ago is "all but the youngest 3 loads + 4 x stores-per-tile" only if that number is known.
template <int ML, int KIND>
__device__ __forceinline__ uint32_t emit_window(WaveOut& W, const uint64_t (&q)[4], const uint32_t (&l)[4], int lane,
                                                uint32_t* seg_dst, uint32_t seg_base, uint64_t* blk_dst, uint64_t blk_val) {
  constexpr int NP = WinGeom<ML>::NP;
  const uint32_t L0 = l[0] + l[1];
  const uint32_t T = L0 + l[2] + l[3];
  const uint32_t incl = wave_incl_scan_u32(T);
  const uint32_t total = wave_last_u32(incl);
  const uint32_t p = W.carry + (incl - T);  // my first bit, relative to st[0]
  // ---- the run, left-justified
  uint32_t P0[NP], P1[NP];
  cat_left64<ML>(q[0] << ((64u - l[0]) & 63u), l[0], q[1] << ((64u - l[1]) & 63u), P0);
  cat_left64<ML>(q[2] << ((64u - l[2]) & 63u), l[2], q[3] << ((64u - l[3]) & 63u), P1);
  // ---- ... each half at its bit phase, NP + 1 dwords of the staging area from its first dword on
  const uint32_t ph[2] = {p, p + L0};
  const uint32_t lh[2] = {L0, T - L0};
  uint32_t rr[2][NP + 1], sh[2];
  uint32_t* w[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t(&Ph)[NP] = h ? P1 : P0;
    sh[h] = ph[h] & 31u;
    rr[h][0] = Ph[0] >> sh[h];
#pragma unroll
    for (int k = 1; k < NP; ++k) rr[h][k] = alignbit(Ph[k - 1], Ph[k], sh[h]);
    rr[h][NP] = alignbit(Ph[NP - 1], 0u, sh[h]);
    w[h] = W.st + (ph[h] >> 5);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t nw = (sh[h] + lh[h] + 31u) >> 5;  // dwords touched (0: no bits and on a boundary)
#pragma unroll
    for (int k = 1; k <= NP; ++k)
      if ((uint32_t)k < nw) w[h][k] = rr[h][k];
    if (sh[h] == 0u && lh[h] != 0u) w[h][0] = rr[h][0];
  }
  wave_sync();
#pragma unroll
  for (int h = 0; h < 2; ++h)
    if (sh[h] != 0u) atomicOr(w[h], rr[h][0]);  // (no bits: ORs nothing)
  wave_sync();
  const uint32_t endbits = W.carry + total;
  const uint32_t U = endbits >> 7;  // <= 128
  if constexpr (KIND == 0) {
    if (seg_dst) *seg_dst = seg_base + incl;  // side-car: where this lane's segment ends, relative to its block
    if (blk_dst) *blk_dst = blk_val;          // side-car: where the block begins (one lane, every fourth tile)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t j = (uint32_t)lane + 64u * h;
      if (j < U) {
        uint4 v = reinterpret_cast<const uint4*>(W.st)[j];
        v.x = bswap32(v.x); v.y = bswap32(v.y); v.z = bswap32(v.z); v.w = bswap32(v.w);
        W.out_units[W.unit_base + j] = v;
      }
    }
  } else {
    if constexpr (KIND == 1) {
      // the segment's end is the inclusive sum of its fourth lane: all four lanes store it (quad_perm [3,3,3,3])
      const uint32_t seg_end = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0xFF, 0xF, 0xF, false);
      *seg_dst = seg_base + seg_end;
      if (blk_dst) *blk_dst = blk_val;  // (compile-time: the first tile of a block; every lane, one address)
    }
    // a full tile holds >= 1024 bits: U >= 8; lanes behind the last complete unit repeat it
    constexpr int NH = ML > 8 ? 2 : 1;  // <= 8-bit codes: at most 127 + 8192 bits, 64 units
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      uint32_t j = (uint32_t)lane + 64u * h;
      j = j < U ? j : U - 1u;
      uint4 v = reinterpret_cast<const uint4*>(W.st)[j];
      v.x = bswap32(v.x); v.y = bswap32(v.y); v.z = bswap32(v.z); v.w = bswap32(v.w);
      W.out_units[W.unit_base + j] = v;
    }
  }
  wave_sync();
  if (KIND != 0 || U) {  // the incomplete unit moves to the front (its partial dword: bits, then zeros; behind it: don't care)
    if (lane < 4) {
      const uint32_t t = W.st[4 * U + lane];
      W.st[lane] = t;
    }
    W.unit_base += U;
    wave_sync();
  }
  W.carry = endbits & 127u;
  return total;
}

// ---- table entries ----
// codes <= 16 bits: u32 = len << 16 | code, 32 replicas; a lane's four-symbol item comes from one input dword.
// Two steps, so that the main loop can re-load the tile register between them: once the sixteen table addresses are
// formed the input bytes are dead, and the next tile can land in the SAME registers (a reload issued while the old
// bytes are still needed costs a second register set and, at the loop's back edge, copies behind a full vmcnt(0)).
__device__ __forceinline__ void lookup_narrow(const uint32_t* tab, uint32_t r, const uint4& v, uint32_t (&e)[16]) {
  const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t b = (vv[k] >> (8 * j)) & 0xFFu;
      e[4 * k + j] = tab[(b << 5) | r];
    }
  }
}
__device__ __forceinline__ void combine_narrow(uint32_t (&e)[16], uint32_t cnt, uint64_t (&q)[4], uint32_t (&l)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if ((uint32_t)(4 * k + j) >= cnt) e[4 * k + j] = 0;  // folds away for cnt == 16
    const uint32_t* ek = e + 4 * k;
    const uint32_t l1 = ek[1] >> 16, l3 = ek[3] >> 16;
    const uint32_t p0 = ((ek[0] & 0xFFFFu) << l1) | (ek[1] & 0xFFFFu);
    const uint32_t p1 = ((ek[2] & 0xFFFFu) << l3) | (ek[3] & 0xFFFFu);
    const uint32_t lp1 = (ek[2] >> 16) + l3;
    q[k] = ((uint64_t)p0 << lp1) | p1;
    l[k] = (ek[0] >> 16) + l1 + lp1;
  }
}
__device__ __forceinline__ void items_narrow(const uint32_t* tab, uint32_t r, const uint4& v, uint32_t cnt, uint64_t (&q)[4],
                                             uint32_t (&l)[4]) {
  uint32_t e[16];
  lookup_narrow(tab, r, v, e);
  combine_narrow(e, cnt, q, l);
}

// codes up to 32 bits: u64 = len << 32 | code, 16 replicas; two symbols per item
__device__ __forceinline__ void items_wide(const uint64_t* tab, uint32_t r, const uint4& v, uint32_t cnt, uint64_t (&q)[8],
                                           uint32_t (&l)[8]) {
  const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const uint32_t b0 = (vv[p >> 1] >> (16 * (p & 1))) & 0xFFu, b1 = (vv[p >> 1] >> (16 * (p & 1) + 8)) & 0xFFu;
    uint64_t e0 = tab[(b0 << 4) | r], e1 = tab[(b1 << 4) | r];
    if ((uint32_t)(2 * p) >= cnt) e0 = 0;
    if ((uint32_t)(2 * p + 1) >= cnt) e1 = 0;
    const uint32_t l1 = (uint32_t)(e1 >> 32);
    q[p] = ((uint64_t)(uint32_t)e0 << l1) | (uint64_t)(uint32_t)e1;
    l[p] = (uint32_t)(e0 >> 32) + l1;
  }
}

// MODE 0: codes up to 32 bits (u64 table entries, eight two-symbol items per lane, ds_or_b32 deposits);
// MODE 1 / 2 / 3: codes up to 9 / 12 / 16 bits (u32 entries, four four-symbol items, the lane's run as plain stores)
template <int MODE>
struct EmitMode {
  static constexpr int NI = 4;
  static constexpr int ML = MODE == 1 ? 9 : (MODE == 2 ? 12 : 16);
  static __device__ __forceinline__ void items(const uint32_t* tab, int lane, const uint4& v, uint32_t cnt, uint64_t (&q)[4],
                                               uint32_t (&l)[4]) {
    items_narrow(tab, (uint32_t)lane & 31u, v, cnt, q, l);
  }
  static __device__ __forceinline__ void one(const uint32_t* tab, int lane, uint32_t byte, uint64_t& code, uint32_t& len) {
    const uint32_t e = tab[(byte << 5) | ((uint32_t)lane & 31u)];
    code = e & 0xFFFFu;
    len = e >> 16;
  }
};
template <>
struct EmitMode<0> {
  static constexpr int NI = 8;
  static constexpr int ML = 32;
  static __device__ __forceinline__ void items(const uint32_t* tab, int lane, const uint4& v, uint32_t cnt, uint64_t (&q)[8],
                                               uint32_t (&l)[8]) {
    items_wide(reinterpret_cast<const uint64_t*>(tab), (uint32_t)lane & 15u, v, cnt, q, l);
  }
  static __device__ __forceinline__ void one(const uint32_t* tab, int lane, uint32_t byte, uint64_t& code, uint32_t& len) {
    const uint64_t e = reinterpret_cast<const uint64_t*>(tab)[(byte << 4) | ((uint32_t)lane & 15u)];
    code = (uint32_t)e;
    len = (uint32_t)(e >> 32);
  }
};
// MODE 4 (k_emit_long): codes of 33..64 bits -- a .crs tree deeper than 32, which takes more than 3.5 million input bytes of
// Fibonacci-distributed counts (huff_tree.cc:157-170 keeps codes as strings of any length).  `tab` is the ghf_code itself, in
// global memory: length[], codeword[] = bits 0..31, symbol[] = bits 32..63 (k_crs_build_code).  One symbol per item.
template <>
struct EmitMode<4> {
  static constexpr int NI = 16;
  static constexpr int ML = 64;
  static __device__ __forceinline__ void one(const uint32_t* tab, int lane, uint32_t byte, uint64_t& code, uint32_t& len) {
    (void)lane;
    len = tab[byte];
    code = ((uint64_t)tab[2 * GHF_NSYM + byte] << 32) | tab[GHF_NSYM + byte];
  }
  static __device__ __forceinline__ void items(const uint32_t* tab, int lane, const uint4& v, uint32_t cnt, uint64_t (&q)[16],
                                               uint32_t (&l)[16]) {
    const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      one(tab, lane, (vv[i >> 2] >> (8 * (i & 3))) & 0xFFu, q[i], l[i]);
      if ((uint32_t)i >= cnt) {
        q[i] = 0;
        l[i] = 0;
      }
    }
  }
};

// A tile's items -> output.  Codes <= 16 bits: one pass (<= 127 + 1024 * 16 bits fit the staging area).  Longer codes:
// the two half-waves one after the other, each at most 32 lanes x 512 bits.
template <int MODE, int KIND>
__device__ __forceinline__ uint32_t emit_tile(WaveOut& W, const uint64_t (&q)[EmitMode<MODE>::NI],
                                              const uint32_t (&l)[EmitMode<MODE>::NI], int lane, uint32_t* seg_out,
                                              bool seg_valid, uint32_t relbits, uint64_t* blk_dst, uint64_t blk_val) {
  constexpr int NI = EmitMode<MODE>::NI;
  if constexpr (MODE == 4) {  // 16 lanes x 16 items x 64 bits per pass
    uint32_t* const seg_dst = (seg_out && seg_valid && (lane & 3) == 3) ? seg_out : nullptr;
    uint32_t total = 0;
#pragma unroll 1
    for (int h = 0; h < 4; ++h) {
      const bool mine = (lane >> 4) == h;
      total += emit_items<NI, false>(W, q, l, mine, lane, mine ? seg_dst : nullptr, relbits + total, h == 0 ? blk_dst : nullptr, blk_val);
    }
    return total;
  } else if constexpr (MODE != 0) {
    uint32_t* seg_dst = seg_out;
    if (KIND == 0) seg_dst = (seg_out && seg_valid && (lane & 3) == 3) ? seg_out : nullptr;  // stored with the tile's units
    return emit_window<EmitMode<MODE>::ML, KIND>(W, q, l, lane, seg_dst, relbits, blk_dst, blk_val);
  } else {
    uint32_t* const seg_dst = (seg_out && seg_valid && (lane & 3) == 3) ? seg_out : nullptr;
    uint32_t total = 0;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      const bool mine = (lane >> 5) == h;
      total += emit_items<NI, false>(W, q, l, mine, lane, mine ? seg_dst : nullptr, relbits + total, h == 0 ? blk_dst : nullptr, blk_val);
    }
    return total;
  }
}

struct EmitGeom {      // wave-uniform facts about the stream
  uint64_t start_bit;  // stream bit of this buffer's first code
  uint64_t origin_byte;  // stream byte that d_out[0] stands for (GHF_EMIT_REBASE), else 0
  int max_len;
};

constexpr int kEmitNB = 4;  // tile buffers = tiles per trip of the main loop = one side-car block

// Everything a chunk's start needs from memory, requested at once, oldest first -- the chunk's bit offset, the two symbols
// per lane in front of it, its first two tiles -- and requested at KERNEL ENTRY, in front of the status check, the code
// tables' trip into LDS and their barriers: none of it depends on them (kernel arguments only), and every workgroup of a
// 256 MiB launch is in the launch's first round, so the five dependent round trips of the old prologue (status, max_len,
// table words, chunk offset, first tiles) were 5 % of the launch for every wave.
constexpr int kEmitPreNB = 2;  // tiles requested at kernel entry (all four would cost the packers two spilled registers)
struct EmitPre {
  uint64_t chunk_off;
  uint32_t prev0, prev1;
  uint4 buf[kEmitPreNB];
};
__device__ __forceinline__ void emit_prefetch(const EmitParams& P, uint32_t c, int lane, EmitPre& R) {
  R.chunk_off = 0;
  R.prev0 = R.prev1 = 0;
#pragma unroll
  for (int j = 0; j < kEmitPreNB; ++j) R.buf[j] = make_uint4(0, 0, 0, 0);
  if (c >= P.nchunks) return;
  const uint64_t chunk = P.chunk;
  const uint64_t sym0 = (uint64_t)c * chunk;
  const uint64_t nsym = (P.n - sym0 < chunk) ? (P.n - sym0) : chunk;
  const uint8_t* pin = P.in + sym0;
  R.chunk_off = P.chunk_off[c];
  if (c > 0) {
    // the previous chunk's last 128 symbols (it is a full chunk, >= 4096 symbols), two per lane: a code has at least
    // one bit, so they cover the <= 127 bits in front of this chunk that share its first unit
    const uint8_t* pp = pin - 128 + 2 * lane;
    R.prev0 = pp[0];
    R.prev1 = pp[1];
  }
  if ((((uintptr_t)pin) & 15u) == 0 && nsym / kSymPerIter >= (uint64_t)kEmitNB) {
    const uint4* pv = reinterpret_cast<const uint4*>(pin) + lane;
#pragma unroll
    for (int j = 0; j < kEmitPreNB; ++j) R.buf[j] = pv[j * 64];
  }
}

template <int MODE>
__device__ __forceinline__ void emit_chunk(const EmitParams& P, const EmitGeom& G, uint32_t c, const uint32_t* tab,
                                           uint32_t* flat, uint32_t* st, int lane, const EmitPre& pre) {
  typedef EmitMode<MODE> M;
  constexpr bool WIDE = MODE == 0 || MODE == 4;  // the deposit-by-OR paths
  constexpr int NI = M::NI;
  const uint64_t chunk = P.chunk;
  const uint64_t sym0 = (uint64_t)c * chunk;
  const uint64_t nsym = (P.n - sym0 < chunk) ? (P.n - sym0) : chunk;
  const uint8_t* pin = P.in + sym0;
  const bool aligned = (((uintptr_t)pin) & 15u) == 0;
  const uint64_t chunk_off = pre.chunk_off;
  const uint32_t prev0 = pre.prev0, prev1 = pre.prev1;
  const uint64_t nfull = aligned ? (nsym / kSymPerIter) : 0;
  const uint4* pv = reinterpret_cast<const uint4*>(pin) + lane;
  // codes beyond 16 bits take the plain loop below (rare, and it keeps that instantiation free of spills); a side-car is
  // written whole or not at all
  constexpr int NB = kEmitNB;
  const bool streamed = !WIDE && nfull >= (uint64_t)NB && ((P.seg_bit != nullptr) == (P.chunk_bit != nullptr));
  uint4 buf[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) buf[j] = j < kEmitPreNB ? pre.buf[j] : make_uint4(0, 0, 0, 0);  // (loaded whenever the chunk has four full, aligned tiles; only `streamed` reads them)
  if (streamed) {
#pragma unroll
    for (int j = kEmitPreNB; j < NB; ++j) buf[j] = pv[j * 64];
  }
  const uint64_t Pc = G.start_bit + chunk_off;
  WaveOut W;
  W.flat = flat;
  W.st = st;
  W.bit0 = (uint32_t)(st - flat) * 32u;
  W.out_units = reinterpret_cast<uint4*>(P.out);
  W.unit_base = (Pc >> 7) - (G.origin_byte >> 4);
  W.carry = (uint32_t)(Pc & 127u);
  for (int i = lane; i < kStageWords / 4; i += 64) reinterpret_cast<uint4*>(st)[i] = make_uint4(0, 0, 0, 0);
  wave_sync();

  // ---- the bits in front of this chunk's first code that share its first 16-byte unit
  if (W.carry) {
    if (c > 0) {
      uint64_t c0, c1;
      uint32_t l0, l1;
      M::one(tab, lane, prev0, c0, l0);
      M::one(tab, lane, prev1, c1, l1);
      const uint32_t lp = l0 + l1;
      const uint32_t incl = wave_incl_scan_u32(lp);
      const uint32_t after = wave_last_u32(incl) - incl;     // bits of the lanes behind me
      int end = (int)W.carry - (int)after;                   // my second symbol ends here (staging bits)
#pragma unroll
      for (int k = 1; k >= 0; --k) {                         // the second symbol, then the first in front of it
        uint64_t cc = k ? c1 : c0;
        uint32_t ll = k ? l1 : l0;
        int start = end - (int)ll;
        if (end > 0 && ll) {
          if (start < 0) {  // only the code's last `end` bits are inside the unit
            ll = (uint32_t)end;
            cc &= ~0ull >> (64u - ll);
            start = 0;
          }
          deposit64(flat, W.bit0 + (uint32_t)start, cc, ll);
        }
        end -= (int)(k ? l1 : l0);
      }
    } else if (!(P.flags & GHF_EMIT_REBASE)) {
      // chunk 0 of a buffer that starts at stream byte 0: bytes in front of the first code stay (the .crs2 header's
      // last words, or whatever the caller put there); a5 writes them here when it rides along
      if (lane < 4) {
        const int bits_before = (int)W.carry - 32 * lane;
        if (bits_before > 0) {
          const uint64_t gw = (Pc >> 7) * 4 + (uint64_t)lane;
          uint32_t word;
          if (P.flags & GHF_EMIT_HEADER) word = header_word(P.code, (int)gw, G.max_len);
          else word = bswap32(reinterpret_cast<const uint32_t*>(P.out)[gw]);
          st[lane] = bits_before >= 32 ? word : (word & (~0u << (32 - bits_before)));
        }
      }
    }
    wave_sync();
  }
  if (c == 0 && (P.flags & GHF_EMIT_HEADER)) {
    // a5, canonical_huff_encoder.cc:210-242: the header words in front of the first unit that holds code bits
    const int nwords = 1 + GHF_NSYM + 2 + 2 * G.max_len;
    const uint64_t first_unit_word = (Pc >> 7) * 4;
    for (int w = lane; w < nwords && (uint64_t)w < first_unit_word; w += 64)
      reinterpret_cast<uint32_t*>(P.out)[w] = bswap32(header_word(P.code, w, G.max_len));
  }

  uint32_t relbits = 0;   // bits of this chunk emitted so far (a chunk has at most 2^20 symbols of <= 32 bits)
  uint32_t blockrel = 0;  // ... since the side-car block (4 tiles = 4096 symbols = 64 segments) began
  uint64_t* const blockp = P.chunk_bit ? P.chunk_bit + (sym0 / kBlockSymbols) : nullptr;
  const uint64_t bit_origin = Pc - G.origin_byte * 8;  // the chunk's first code, relative to d_out[0]
  auto tile_done = [&](uint64_t it, uint32_t total) {  // wave-uniform bookkeeping after tile `it`
    relbits += total;
    blockrel = ((it & 3) == 3) ? 0u : blockrel + total;
  };
  auto block_dst = [&](uint64_t it) -> uint64_t* {  // stored with the tile's units (after its drain)
    return (blockp && (it & 3) == 0 && lane == 0) ? blockp + (it >> 2) : nullptr;
  };
  uint32_t* const segp = P.seg_bit ? P.seg_bit + ((sym0 + (uint64_t)lane * 16) >> 6) : nullptr;
  uint64_t it = 0;
  // ---- full 1 KiB tiles, four per trip (= one side-car block).  While a tile is packed the loads of the next THREE are in
  //      flight: a buffer is re-loaded as soon as its bytes have become table addresses (same registers, no copy), and every
  //      global store of the trip is an unconditional straight-line instruction, so the compiler's own wait in front of a
  //      buffer's first use is "all but the 3 younger loads and the stores issued since" -- the in-order vmcnt of gfx9
  //      would otherwise force every prefetch to land before the NEXT tile's stores.  The loop is entered with nothing in
  //      flight (one full wait per chunk), so that these waits are computed from its own back edge alone.  The last
  //      prefetches are clamped to the last full tile and simply unused.
  if constexpr (!WIDE) {
    if (streamed) {
      auto trips = [&](auto kind_tag) {
        constexpr int KIND = decltype(kind_tag)::value;
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        for (; it + NB <= nfull; it += NB) {
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            uint32_t e[16];
            lookup_narrow(tab, (uint32_t)lane & 31u, buf[j], e);
            __builtin_amdgcn_sched_barrier(0);  // the bytes of buf[j] are consumed: its registers take the next load
            const uint64_t nx = (it + NB + j < nfull) ? it + NB + j : nfull - 1;
            buf[j] = pv[nx * 64];
            __builtin_amdgcn_sched_barrier(0);
            uint64_t q[NI];
            uint32_t l[NI];
            combine_narrow(e, 16, q, l);
            const uint32_t total = emit_tile<MODE, KIND>(W, q, l, lane, KIND == 1 ? segp + (it + j) * 16 : nullptr, true, blockrel,
                                                         (KIND == 1 && j == 0) ? blockp + (it >> 2) : nullptr, bit_origin + relbits);
            relbits += total;
            blockrel = (j == NB - 1) ? 0u : blockrel + total;
          }
        }
      };
      if (segp) trips(std::integral_constant<int, 1>{});
      else trips(std::integral_constant<int, 2>{});
    }
  }
  // ---- whatever is left: an odd full tile, the ragged tail, unaligned input, and -- on the stream's last chunk --
  //      one extra pass for the end mark and the padding (canonical_huff_encoder.cc:255-257, buffer.h:277-280)
  const uint64_t niter = (nsym + kSymPerIter - 1) / kSymPerIter;
  const bool last = (P.flags & GHF_EMIT_LAST) && c + 1 == P.nchunks;
  const uint64_t nsteps = niter + (last ? 1 : 0);
#pragma unroll 1
  for (; it < nsteps; ++it) {
    uint64_t q[NI];
    uint32_t l[NI];
    bool seg_valid = false;
    uint32_t* seg_out = nullptr;
    uint64_t* blk = nullptr;
    if (it < niter) {
      const uint64_t sb = it * kSymPerIter;
      const uint64_t rem = nsym - sb;
      uint4 v;
      uint32_t cnt = 16;
      if (aligned && rem >= (uint64_t)kSymPerIter) {
        v = pv[it * 64];
      } else {  // byte loads, never past the end of the buffer
        const uint64_t lo = (uint64_t)lane * 16;
        cnt = rem > lo ? (rem - lo >= 16 ? 16u : (uint32_t)(rem - lo)) : 0u;
        uint32_t w[4] = {0, 0, 0, 0};
        for (uint32_t j = 0; j < cnt; ++j) w[j >> 2] |= (uint32_t)pin[sb + lo + j] << (8 * (j & 3));
        v = make_uint4(w[0], w[1], w[2], w[3]);
      }
      M::items(tab, lane, v, cnt, q, l);
      seg_valid = rem > (uint64_t)(lane & ~3) * 16;  // my segment has at least one symbol
      seg_out = segp ? segp + it * 16 : nullptr;
      blk = block_dst(it);
    } else {
      const uint32_t el = P.code->length[GHF_NSYM - 1], ec = P.code->codeword[GHF_NSYM - 1];
      const uint32_t pad = (uint32_t)((0 - (Pc + relbits + el)) & 7u);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        q[i] = 0;
        l[i] = 0;
      }
      if (lane == 0) {
        q[0] = ((uint64_t)ec << pad) | ((1u << pad) - 1u);
        l[0] = el + pad;
      }
    }
    tile_done(it, emit_tile<MODE, 0>(W, q, l, lane, seg_out, seg_valid, blockrel, blk, bit_origin + relbits));
  }
  // the chunk's last, incomplete unit belongs to the next chunk's wave -- unless this is the buffer's last chunk
  // (what lies behind the last bit is stored as zeros: a neighbouring shard or file piece ORs its own bits into this unit)
  if (c + 1 == P.nchunks && W.carry && lane == 0) {
    const uint4 v = *reinterpret_cast<const uint4*>(st);
    uint32_t ww[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int vb = (int)W.carry - 32 * i;  // valid bits of this dword
      ww[i] = vb >= 32 ? ww[i] : (vb <= 0 ? 0u : (ww[i] & (~0u << (32 - vb))));
    }
    W.out_units[W.unit_base] = make_uint4(bswap32(ww[0]), bswap32(ww[1]), bswap32(ww[2]), bswap32(ww[3]));
  }
}

// what both kernels start with: a failed earlier stage or an output that does not fit -> uniform exit; where the stream ends.
// `mine` = this kernel is the one that packs codes of this length (the other one of the pair leaves without a trace).
__device__ __forceinline__ bool emit_begin(const EmitParams& P, EmitGeom& G, int* status0, bool long_kernel) {
  const int tid = threadIdx.x;
  if (tid == 0) *status0 = *P.status;  // a previous stage failed -> uniform exit (checked below: its round trip overlaps the others)
  G.max_len = P.code->max_len;
  G.start_bit = P.d_start_bit ? *P.d_start_bit : 8ull * (1040ull + 8ull * (uint64_t)G.max_len);
  G.origin_byte = (P.flags & GHF_EMIT_REBASE) ? ((G.start_bit >> 7) << 4) : 0ull;
  // where the stream ends; does it fit?  (every workgroup computes the same answer from the same few words)
  uint64_t end = G.start_bit + P.chunk_off[P.nchunks];
  if (P.flags & GHF_EMIT_LAST) {
    end += P.code->length[GHF_NSYM - 1];
    end = (end + 7) & ~7ull;
  }
  const uint64_t end_byte = ((end + 7) >> 3) - G.origin_byte;
  const uint64_t end_unit_bytes = (((end >> 7) + 1) << 4) - G.origin_byte;  // through the unit holding the end bit
  const bool fits = end_unit_bytes <= P.cap || (((end & 127u) == 0) && end_byte <= P.cap);
  __syncthreads();
  if (*status0 != 0) return false;  // (what was read above is then meaningless, and unused)
  if ((G.max_len > 32) != long_kernel) return false;
  if (blockIdx.x == 0 && tid == 0) {
    if (!fits) latch_status(P.status, GHF_E_CAP);
    if (P.d_end) {
      P.d_end[0] = end;
      P.d_end[1] = end_byte;
    }
  }
  return fits;
}

__global__ __launch_bounds__(kEmitThreads, 6) void k_emit(EmitParams P) {
  __shared__ __attribute__((aligned(16))) uint32_t tab[kEmitTabWords];
  __shared__ __attribute__((aligned(16))) uint32_t stage[kEmitWaves * kStageWords];
  __shared__ int status0;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // tell the compiler it is wave-uniform (SGPR)
  const uint32_t c = blockIdx.x * kEmitWaves + wave;
  // ---- every load of the prologue goes out here, before the first barrier: this wave's chunk start, this thread's table words
  EmitPre pre;
  emit_prefetch(P, c, lane, pre);
  constexpr int kTabTrips = 1024 / kEmitThreads;
  static_assert(1024 % kEmitThreads == 0, "table slots per thread");
  uint32_t tcode[kTabTrips], tlen[kTabTrips];
#pragma unroll
  for (int k = 0; k < kTabTrips; ++k) {
    const int sy = (tid + k * kEmitThreads) >> 2;
    tcode[k] = P.code->codeword[sy];
    tlen[k] = P.code->length[sy];
  }
  EmitGeom G;
  if (!emit_begin(P, G, &status0, false)) {
    // codes beyond 32 bits are k_emit_long's (queued behind this kernel when the caller said GHF_EMIT_LONG_CODES);
    // without that flag nobody would pack them
    if (status0 == 0 && P.code->max_len > 32 && !(P.flags & GHF_EMIT_LONG_CODES) && blockIdx.x == 0 && tid == 0)
      latch_status(P.status, GHF_E_FORMAT);
    return;
  }
  const bool wide = G.max_len > 16;
  {
    // 4 slots per symbol, each 32 bytes of the symbol's 128-byte row: replicas 8 * (i & 3) .. (u32) or 4 * (i & 3) .. (u64).
    // The tables may be the caller's own: the packers below size their registers and the staging area by max_len.
    bool bad = G.max_len < 1;
#pragma unroll
    for (int k = 0; k < kTabTrips; ++k) {
      const int i = tid + k * kEmitThreads;
      const int s = i >> 2;
      const uint32_t code = tcode[k], len = tlen[k];
      bad |= len > (uint32_t)G.max_len;
      uint4 v;
      if (!wide) {
        const uint32_t e = (len << 16) | (code & 0xFFFFu);
        v = make_uint4(e, e, e, e);
      } else {
        v = make_uint4(code, len, code, len);
      }
      uint4* dst = reinterpret_cast<uint4*>(tab) + (s * 8 + (i & 3) * 2);
      dst[0] = v; dst[1] = v;
    }
    if (__syncthreads_or(bad)) {
      if (tid == 0) latch_status(P.status, GHF_E_FORMAT);
      return;
    }
  }
  __syncthreads();
  if (c >= P.nchunks) return;
  uint32_t* const st = stage + wave * kStageWords;
  if (G.max_len <= 9) emit_chunk<1>(P, G, c, tab, stage, st, lane, pre);
  else if (G.max_len <= 12) emit_chunk<2>(P, G, c, tab, stage, st, lane, pre);
  else if (!wide) emit_chunk<3>(P, G, c, tab, stage, st, lane, pre);
  else emit_chunk<0>(P, G, c, tab, stage, st, lane, pre);
}

// Codes of 33..64 bits (a .crs whose tree is deeper than 32).  Same chunks, same one-writer-per-unit layout, same
// side-car; the tables are read from global memory (three L2-resident words per symbol) and every symbol is an item of its
// own.  Rare and slow by design: it exists so that every stream the reference can write can be written here.
__global__ __launch_bounds__(kEmitThreads) void k_emit_long(EmitParams P) {
  __shared__ __attribute__((aligned(16))) uint32_t stage[kEmitWaves * kStageWords];
  __shared__ int status0;
  const int tid = threadIdx.x;
  EmitGeom G;
  if (!emit_begin(P, G, &status0, true)) return;
  bool bad = G.max_len > 64 || (P.flags & GHF_EMIT_LAST) != 0;  // (no end mark in this format; 64 + 7 bits would not fit an item)
  for (int s = tid; s < 256; s += kEmitThreads) bad |= P.code->length[s] > (uint32_t)G.max_len;
  if (__syncthreads_or(bad)) {
    if (tid == 0) latch_status(P.status, GHF_E_FORMAT);
    return;
  }
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t c = blockIdx.x * kEmitWaves + wave;
  if (c >= P.nchunks) return;
  EmitPre pre;
  emit_prefetch(P, c, lane, pre);
  emit_chunk<4>(P, G, c, reinterpret_cast<const uint32_t*>(P.code), stage, stage + wave * kStageWords, lane, pre);
}

void launch_emit(const EmitParams& p, hipStream_t s) {
  const uint32_t blocks = (p.nchunks + kEmitWaves - 1) / kEmitWaves;
  if (blocks == 0) return;
  hipLaunchKernelGGL(k_emit, dim3(blocks), dim3(kEmitThreads), 0, s, p);
  if (p.flags & GHF_EMIT_LONG_CODES) hipLaunchKernelGGL(k_emit_long, dim3(blocks), dim3(kEmitThreads), 0, s, p);
}

}  // namespace ghf
