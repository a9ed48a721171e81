// golden-huffman_amd/csrc/ghf_decode.hip -- K7 (table-driven block-parallel decode) and K6 (side-car reconstruction
// for streams that come without one), gfx950 / wave64.  File:line citations are relative to the reference tree.
#include "ghf_device.h"

namespace ghf {

// ------------------------------------------------------------------------------------------------
// K7: decode.  k_build_decode_tables turns the header tables into left-justified first codes
// (FastCanonicalHuffDecoder, canonical_huff_encoder.cc:433-434) and a 2^lut_bits direct table
// {symbol, length} -- the reference's 8-bit length LUT (canonical_huff_encoder.cc:466-516) widened
// to min(max_len, 12) bits so that no linear extension is needed at any BASELINE config; longer
// codes fall back to the reference's linear search over first_code (cfind, canonical_huff_encoder.h:157-162).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kEntEnd = 1u << 16, kEntNone = 1u << 17;
__device__ __forceinline__ uint32_t dec7_entry(uint32_t g) {  // compact entry (sym | len << 9) -> image entry
  const uint32_t sym = g & 0x1FFu, len = g >> 9;
  return (sym & 0xFFu) | (len << 8) | (sym == 256u ? kEntEnd : 0u) | (len == 0u ? kEntNone : 0u);
}
// DecTables::image from the compact tables (LDS): every table has at least four copies of an entry side by side, so the
// image is written 16 bytes at a time
__device__ __forceinline__ void dec_image_fill(DecTables* __restrict__ dt, const uint16_t* lut, const uint32_t* lut2, int lb, int pb, int tid,
                                               int nthreads) {
  uint4* const img4 = reinterpret_cast<uint4*>(dt->image);
  if (pb) {
    const int r2 = (kDec7LutLog2 - pb) < 5 ? (kDec7LutLog2 - pb) : 5;
    for (int g = tid; g < (1 << (pb + r2 - 2)); g += nthreads) {
      const uint32_t e = lut2[(4 * g) >> r2];
      img4[g] = make_uint4(e, e, e, e);
    }
    for (int g = tid; g < kDec7SmallSlots / 4; g += nthreads) {
      const uint32_t e = dec7_entry(lut[((4 * g) >> 5) & ((1 << lb) - 1)]);
      img4[kDec7LutSlots / 4 + g] = make_uint4(e, e, e, e);
    }
  } else {
    const int r1 = (kDec7LutLog2 - lb) < 5 ? (kDec7LutLog2 - lb) : 5;
    for (int g = tid; g < (1 << (lb + r1 - 2)); g += nthreads) {
      const uint32_t e = dec7_entry(lut[(4 * g) >> r1]);
      img4[g] = make_uint4(e, e, e, e);
    }
  }
}

__global__ __launch_bounds__(256) void k_build_decode_tables(const ghf_code* __restrict__ code, DecTables* __restrict__ dt,
                                                             int* __restrict__ status) {
  __shared__ uint32_t fcl[36];
  __shared__ uint32_t sp[36];
  __shared__ unsigned long long kraft;
  __shared__ int bad;
  __shared__ uint16_t s_lut[1 << kDecLutBitsMax];    // sym | len << 9 ; 0 = code longer than lut_bits
  __shared__ uint32_t s_lut2[1 << kDecPairBitsMax];  // sym0 | sym1 << 8 | (len0 + len1) << 16 ; bit 30 = not two data symbols
  const int tid = threadIdx.x;
  const int max_len = code->max_len, min_len = code->min_len;
  if (max_len < 1 || max_len > 32 || min_len < 1 || min_len > max_len) {
    if (tid == 0) latch_status(status, GHF_E_FORMAT);
    return;
  }
  // The tables may come from anywhere (ghf_parse_header checks a header on the host; a caller's own ghf_code is not
  // checked by anyone else).  A Huffman code over >= 2 symbols is COMPLETE: the lengths satisfy Kraft with equality
  // (sum of 2^-len == 1), every length lies in [min_len, max_len], first codes fit their length and start positions
  // stay inside symbol[].  Anything else would leave table entries without a code (length 0) and is refused.
  if (tid == 0) {
    kraft = 0;
    bad = 0;
  }
  __syncthreads();
  {
    unsigned long long k = 0;
    int b = 0;
    for (int i = tid; i < GHF_NSYM; i += 256) {
      const uint32_t l = code->length[i];
      if (l) {
        if ((int)l < min_len || (int)l > max_len) b = 1;
        else k += 1ull << (32 - l);
      }
    }
    if (tid >= min_len && tid <= max_len) {
      const uint32_t fc = code->first_code[tid];
      if ((tid < 32 && fc > (1u << tid)) || code->start_pos[tid] > (uint32_t)GHF_NSYM) b = 1;
    }
    if (k) atomicAdd(&kraft, k);
    if (b) atomicOr(&bad, 1);
  }
  __syncthreads();
  const bool lone_end_mark = max_len == 1 && kraft == (1ull << 31) && code->length[GHF_NSYM - 1] == 1;  // GHF_EMPTY_OK's stream
  if (bad || (kraft != (1ull << 32) && !lone_end_mark)) {
    if (tid == 0) latch_status(status, GHF_E_FORMAT);
    return;
  }
  const int lb = max_len < kDecLutBitsMax ? max_len : kDecLutBitsMax;
  if (tid < 36) {
    uint32_t f = 0xFFFFFFFFu, p = 0;
    if (tid >= min_len && tid <= max_len) {
      f = code->first_code[tid] << (32 - tid);
      p = code->start_pos[tid];
    }
    fcl[tid] = f;
    sp[tid] = p;
    dt->fc_left[tid] = f;
    dt->start_pos[tid] = p;
  }
  for (int i = tid; i < GHF_NSYM; i += 256) dt->symbol[i] = (uint16_t)(code->symbol[i] > 256u ? 256u : code->symbol[i]);
  // two symbols per lookup when any two codes fit the index (small alphabets: 16-symbol data has max_len 5)
  const int pb = 2 * max_len <= kDecPairBitsMax ? 2 * max_len : 0;
  if (tid == 0) {
    dt->min_len = min_len;
    dt->max_len = max_len;
    dt->lut_bits = lb;
    dt->pair_bits = pb;
    dt->kind = 0;
    dt->root = 0;
    dt->done = 0;
  }
  if (tid < 16) dt->ticket[tid * 32] = 0;
  __syncthreads();
  auto one = [&](uint32_t v, int upto) -> uint32_t {  // sym | len << 9 of the code of <= upto bits at the top of v; 0: none
    for (int len = min_len; len <= upto; ++len) {
      if (v >= fcl[len]) {
        const uint32_t k = sp[len] + ((v - fcl[len]) >> (32 - len));
        const uint32_t sym = k < GHF_NSYM ? code->symbol[k] : 256u;
        return (sym > 256u ? 256u : sym) | ((uint32_t)len << 9);
      }
    }
    return 0u;
  };
  for (uint32_t idx = tid; idx < (1u << lb); idx += 256) s_lut[idx] = (uint16_t)one(idx << (32 - lb), lb);
  for (uint32_t idx = tid; pb && idx < (1u << pb); idx += 256) {
    const uint32_t v = idx << (32 - pb);
    const uint32_t e0 = one(v, max_len);
    uint32_t ent = (1u << 30) | (1u << 16);  // not a data symbol: flagged, one bit consumed
    if (e0 && (e0 & 0x1FFu) != 256u) {
      const uint32_t l0 = e0 >> 9;
      const uint32_t e1 = one(v << l0, max_len);
      if (e1 && (e1 & 0x1FFu) != 256u) ent = (e0 & 0xFFu) | ((e1 & 0xFFu) << 8) | ((l0 + (e1 >> 9)) << 16);
      else ent = (1u << 30) | (l0 << 16);
    }
    s_lut2[idx] = ent;
  }
  __syncthreads();
  dec_image_fill(dt, s_lut, s_lut2, lb, pb, tid, 256);
}

// .crs: the same direct table, filled by walking the tree DecodeHuffTree::do_build_tree would rebuild
// (include/huff_tree.cc:289-303); what the table cannot resolve is walked bit by bit like decode_byte does (:255-271).
__global__ __launch_bounds__(256) void k_crs_decode_tables(const ghf_tree* __restrict__ tree, DecTables* __restrict__ dt,
                                                           int* __restrict__ status) {
  __shared__ uint16_t tl[256], tr[256];
  __shared__ uint32_t s_min;
  __shared__ uint16_t s_lut[1 << kDecLutBitsMax];
  const int tid = threadIdx.x;
  const int max_len = (int)tree->max_len;
  const uint32_t root = tree->root, nl = tree->n_leaves;
  if (max_len < 1 || max_len > 64 || nl < 2 || nl > 256 || root < 256 || root >= 256 + nl - 1) {
    if (tid == 0) latch_status(status, GHF_E_FORMAT);
    return;
  }
  tl[tid] = tree->left[tid];
  tr[tid] = tree->right[tid];
  if (tid == 0) s_min = 64;
  __syncthreads();
  const int lb = max_len < kDecLutBitsMax ? max_len : kDecLutBitsMax;
  dt->tl[tid] = tl[tid];
  dt->tr[tid] = tr[tid];
  if (tid < 36) {
    dt->fc_left[tid] = 0xFFFFFFFFu;
    dt->start_pos[tid] = 0;
  }
  for (int i = tid; i < GHF_NSYM; i += 256) dt->symbol[i] = 256;
  if (tid < 16) dt->ticket[tid * 32] = 0;
  uint32_t mn = 64;
  for (uint32_t idx = tid; idx < (1u << lb); idx += 256) {
    uint32_t node = root;
    uint16_t ent = 0;
    for (int l = 1; l <= lb; ++l) {
      const uint32_t p = node - 256u;
      if (p >= nl - 1) break;  // a child id that is neither a leaf nor one of the nl - 1 parents: malformed, entry stays 0
      node = ((idx >> (lb - l)) & 1u) ? tr[p] : tl[p];
      if (node < 256u) {
        ent = (uint16_t)(node | ((uint32_t)l << 9));
        mn = (uint32_t)l < mn ? (uint32_t)l : mn;
        break;
      }
    }
    s_lut[idx] = ent;
  }
  atomicMin(&s_min, mn);
  __syncthreads();
  if (tid == 0) {
    dt->min_len = (int32_t)(s_min <= (uint32_t)lb ? s_min : (uint32_t)lb);
    dt->max_len = max_len;
    dt->lut_bits = lb;
    dt->pair_bits = 0;
    dt->kind = 1;
    dt->root = root;
    dt->done = 0;
  }
  dec_image_fill(dt, s_lut, nullptr, lb, 0, tid, 256);
}

void launch_crs_decode_tables(const ghf_tree* d_tree, DecTables* d_dt, int* d_status, hipStream_t s) {
  hipLaunchKernelGGL(k_crs_decode_tables, dim3(1), dim3(256), 0, s, d_tree, d_dt, d_status);
}

void launch_build_decode_tables(const ghf_code* d_code, DecTables* d_dt, int* d_status, hipStream_t s) {
  hipLaunchKernelGGL(k_build_decode_tables, dim3(1), dim3(256), 0, s, d_code, d_dt, d_status);
}

// K7 keeps the direct table in LDS as 32-bit entries, REPLICATED so that the 64 random lookups of a wave do not pile
// up on a few banks: the table gets 64 KiB = 16384 slots; with lut_bits index bits there is room for
// R = min(32, 2^(14 - lut_bits)) copies, slot = index * R + lane % R.  Up to 9-bit tables (uniform bytes: 8/9-bit codes)
// that is one bank per lane of a 32-lane LDS group: conflict-free whatever the data (PMC, 256 MiB uniform, round 1:
// 74 % of the LDS cycles of the 16-bit / 16-copy layout were bank-conflict cycles).  12-bit tables get 4 copies (skewed
// data hits few, mostly identical entries anyway: identical addresses broadcast).
//   entry = symbol | length << 8 | bit 16: end mark | bit 17: no code of <= lut_bits bits starts with these bits
// The room comes from the output: a lane keeps its 64 decoded bytes in 16 registers and the wave's INPUT tile, dead by
// then, serves as the transposition buffer for the coalesced copy-out.
constexpr int kDec7Threads = 1024;
constexpr int kDec7Waves = kDec7Threads / kWave;
constexpr int kDec7InBytes = 4608;  // staged span per wave: 4096 symbols at <= 9 bits average (a byte-Huffman code averages <= 8.1)
constexpr int kDec7InWords = kDec7InBytes / 4;
// The input tiles are PADDED: 16 bytes after every 128.  A lane's segment of uniform bytes is ~64 bytes long, so the 32
// lanes of an LDS group read "their current word" 16 words apart -- two banks for 32 lanes, a 16-way conflict on every
// window refill (PMC, round 2: 68 % of this kernel's LDS cycles).  With the pad, lanes two apart shift by four banks and
// only lanes l, l + 16 still share one (2-way: free).  All tiles live in one logical byte space (tile stride a multiple
// of 128) so that logical -> physical is two VALU instructions, no per-wave base: phys = la + (la >> 7 << 4).
constexpr int kDec7TileLog = kDec7InBytes + 128;                  // logical bytes per wave (16 zero bytes + slack behind the span)
constexpr int kDec7TilePhys = kDec7TileLog / 128 * 144;           // 5328
static_assert(kDec7TileLog % 128 == 0 && kDec7TilePhys >= 4096 + 16, "tile doubles as the 4 KiB transposition buffer");
__device__ __forceinline__ uint32_t in_phys(uint32_t la) {
  // two instructions, v_lshrrev + v_lshl_add (left to itself the compiler canonicalises (la >> 7) << 4 into shift, mask, add:
  // three -- and this sits in every window refill of every decoder)
  uint32_t t = la >> 7;
  asm("" : "+v"(t));
  return (t << 4) + la;
}
struct DecLds7 {
  alignas(128) uint8_t in[kDec7Waves * kDec7TilePhys];  // compressed spans of the waves' groups, big-endian words, padded; then their output
  alignas(16) uint32_t lut[kDec7LutSlots + kDec7SmallSlots];
  uint32_t fcl[36];
  uint32_t sp[36];
  uint16_t symbol[GHF_NSYM + 3];
  uint16_t tl[256], tr[256];  // kind 1 (.crs): the tree
  uint32_t root;
  int kind;
  int status0;
};
static_assert(sizeof(DecLds7) <= 160 * 1024, "one workgroup of 16 waves per CU");
template <typename LT>
__device__ __forceinline__ void dec_small_load(LT& L, const DecTables* dt, int tid, int nthreads) {
  if (tid < 36) {
    L.fcl[tid] = dt->fc_left[tid];
    L.sp[tid] = dt->start_pos[tid];
  }
  for (int i = tid; i < GHF_NSYM; i += nthreads) L.symbol[i] = dt->symbol[i];
  for (int i = tid; i < 256; i += nthreads) {
    L.tl[i] = dt->tl[i];
    L.tr[i] = dt->tr[i];
  }
  if (tid == 0) {
    L.kind = dt->kind;
    L.root = dt->root;
  }
}

// the table image (DecTables::image, written by the table kernels in its final layout) and the small tables into LDS
__device__ __forceinline__ void dec_lds_load7(DecLds7& L, const DecTables* dt, int tid, int nthreads) {
  constexpr int kVecs = (kDec7LutSlots + kDec7SmallSlots) / 4;
  const uint4* const img4 = reinterpret_cast<const uint4*>(dt->image);
  uint4* const lut4 = reinterpret_cast<uint4*>(L.lut);
  for (int g = tid; g < kVecs; g += nthreads) lut4[g] = img4[g];
  dec_small_load(L, dt, tid, nthreads);
}

// codes longer than the direct table: the reference's linear extension (canonical_huff_encoder.cc:554-557).
// returns sym | len << 16
template <typename LT>
__device__ __forceinline__ uint32_t dec_long(const LT& L, uint32_t hi, int lut_bits, int max_len) {
  if (L.kind == 1) {  // .crs: walk the tree from the root (huff_tree.cc:255-271); malformed trees end in "no symbol"
    uint32_t node = L.root;
    for (int l = 1; l <= max_len && l <= 32; ++l) {
      const uint32_t p = node - 256u;
      if (p >= 256u) break;
      node = ((hi >> (32 - l)) & 1u) ? L.tr[p] : L.tl[p];
      if (node < 256u) return node | ((uint32_t)l << 16);
    }
    return 256u | ((uint32_t)max_len << 16);
  }
  int l = lut_bits + 1;
  if (l > max_len) return 256u | ((uint32_t)max_len << 16);  // an incomplete table (bits no code starts with): no symbol
  while (l < max_len && hi < L.fcl[l]) ++l;
  const uint32_t k = L.sp[l] + ((hi - L.fcl[l]) >> (32 - l));
  return (k < GHF_NSYM ? (uint32_t)L.symbol[k] : 256u) | ((uint32_t)l << 16);
}
// the same walk over 64 stream bits: a .crs tree deeper than 32 (include/huff_tree.cc:157-170 keeps codes as strings; such a
// tree needs more than 3.5 million input bytes).  The callers' windows hold at least 65 bits behind the cursor.
template <typename LT>
__device__ __forceinline__ uint32_t dec_long64(const LT& L, uint64_t hi, int max_len) {
  uint32_t node = L.root;
  for (int l = 1; l <= max_len; ++l) {
    const uint32_t p = node - 256u;
    if (p >= 256u) break;
    node = ((hi >> (64 - l)) & 1ull) ? L.tr[p] : L.tl[p];
    if (node < 256u) return node | ((uint32_t)l << 16);
  }
  return 256u | ((uint32_t)max_len << 16);
}
// the next 64 stream bits of a cursor {W: 64 bits, o < 32 of them consumed; nextw: the 32 bits behind W}
__device__ __forceinline__ uint64_t window64(uint64_t W, uint32_t nextw, uint32_t o) {
  return o ? ((W << o) | ((uint64_t)nextw >> (32u - o))) : W;
}

// big-endian word at logical byte address la of the padded input tiles
__device__ __forceinline__ uint32_t in_word(const uint8_t* lin, uint32_t la) { return *reinterpret_cast<const uint32_t*>(lin + in_phys(la)); }

template <bool STAGED>
struct DecIn {
  const uint8_t* lin;   // staged: L.in
  uint32_t la0;         // staged: logical byte address of the wave's tile
  const uint8_t* src;   // unstaged: raw bytes of the span
  uint64_t span;
  __device__ __forceinline__ uint32_t fetch(uint32_t widx) const {
    if (STAGED) return in_word(lin, la0 + 4u * widx);
    const uint64_t b = (uint64_t)widx * 4;
    uint32_t r = 0;
    for (int k = 0; k < 4; ++k) r = (r << 8) | (b + k < span ? (uint32_t)src[b + k] : 0u);
    return r;
  }
};

// what a lane needs to look codes up in ITS replica of a table
struct DecLut {
  const char* base;  // table + 4 * (lane % copies)
  int lsh;           // 32 - index bits
  int ash;           // log2(copies) + 2
};
__device__ __forceinline__ uint32_t dec_lookup(const DecLut& T, uint32_t v) {
  return *reinterpret_cast<const uint32_t*>(T.base + ((v >> T.lsh) << T.ash));
}
template <typename LT>
__device__ __forceinline__ uint32_t dec_long_entry(const LT& L, uint32_t v, int lut_bits, int max_len) {
  const uint32_t r = dec_long(L, v, lut_bits, max_len);  // sym | len << 16
  return (r & 0xFFu) | ((r >> 16) << 8) | ((r & 0x100u) << 8);
}
// ... for a cursor whose code may be longer than 32 bits (o < 32: the refill in front of every lookup guarantees it)
template <typename LT>
__device__ __forceinline__ uint32_t dec_long_entry_at(const LT& L, uint64_t W, uint32_t nextw, uint32_t o, int lut_bits, int max_len) {
  if (max_len <= 32) return dec_long_entry(L, (uint32_t)((W << o) >> 32), lut_bits, max_len);
  const uint32_t r = dec_long64(L, window64(W, nextw, o), max_len);
  return (r & 0xFFu) | ((r >> 16) << 8) | ((r & 0x100u) << 8);
}

// The window W holds 64 stream bits, `o` of them (from the top) already consumed; one symbol costs a 64-bit shift, the
// table lookup and an add.  K symbols are decoded between two refill checks -- K * max_len <= 32 keeps o + max_len <= 64
// at every lookup.
#define GHF_REFILL()           \
  if (o >= 32u) {              \
    W = (W << 32) | nextw;     \
    o -= 32u;                  \
    nextw = in_word(lin, la);  \
    la += 4u;                  \
  }
#define GHF_WINDOW_OPEN()                                              \
  uint32_t la = la0 + ((pos >> 5) << 2);                               \
  uint32_t o = pos & 31u;                                              \
  uint64_t W = ((uint64_t)in_word(lin, la) << 32) | in_word(lin, la + 4u); \
  uint32_t nextw = in_word(lin, la + 8u);                              \
  la += 12u
#define GHF_WINDOW_USED() ((la - la0 - 12u) * 8u + o - pos)  // (la - la_first - 12) * 8 + o - o0, la_first and o0 being pos's two halves

// HOT: the 64 symbols of a full, staged segment; the 64 bytes stay in registers.  LONG: codes beyond the direct table exist
// (max_len > 12: Zipf 1.1 over 256 values has 13..14-bit codes for its rarest ones and the end mark at 256 MiB) and take the
// reference's linear extension on a miss.  Returns the OR of all entries (kEntEnd / kEntNone set: not 64 data symbols ->
// corrupt).  Every variant ends in the same four stores (the callers' copy-out), so that the compiler can count the kernel's
// memory operations whichever variant runs.
//
// A LOOP, not 64 unrolled lookups: the straight-line form of round 3 (and of this round's first build) made a pass 8..9 KB of
// code and kept 39 registers more alive.  (Code size itself is NOT what that costs: loop bodies up to 16 KB issue at the full
// rate, scratch/ifetch2.hip, profiles/r04/experiments/ifetch2.txt -- the "cliff at 4 KB" an earlier micro-benchmark showed was a
// macro that repeated twice as often as its name said.)  The body is one period of the refill pattern -- lcm(4, K) symbols,
// at most 12 = about 0.8 KB -- and the decoded dwords go to out[] through the uniform loop counter (s_set_gpr_idx: no
// scratch); with 89 registers a K7 workgroup shares its CU with the one-wave code build of a later step.
typedef uint32_t DecOut __attribute__((ext_vector_type(16)));  // a lane's 64 decoded bytes: a register TUPLE, so that out[t] with a
                                                                // uniform t is an indexed register move and never memory
template <int K, bool LONG, typename LT>
__device__ __forceinline__ uint32_t dec_hot(const LT& L, const uint8_t* lin, uint32_t la0, const DecLut& T, int lut_bits, int max_len, uint32_t pos,
                                            DecOut& out, uint32_t& used) {
  GHF_WINDOW_OPEN();
  uint32_t acc = 0;
  constexpr int PER = (K == 3) ? 3 : 1;  // dwords per trip: the refill checks of a trip sit at the same symbols in every trip
  auto dword = [&](int sym0) -> uint32_t {  // symbols sym0 .. sym0 + 3 of the trip -> one output dword
    uint32_t e[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if ((sym0 + j) % K == 0) GHF_REFILL();
      const uint32_t v = (uint32_t)((W << o) >> 32);
      uint32_t ent = dec_lookup(T, v);
      if (LONG && __builtin_expect((ent & kEntNone) != 0, 0)) ent = dec_long_entry(L, v, lut_bits, max_len);
      o += (ent >> 8) & 0xFFu;
      e[j] = ent;
    }
    acc |= e[0] | e[1] | e[2] | e[3];
    // byte 0 of four entries -> one dword (v_perm_b32): {a.b0, b.b0} then {lo16, hi16}
    const uint32_t lo = __builtin_amdgcn_perm(e[1], e[0], 0x0C0C0400u);
    const uint32_t hi = __builtin_amdgcn_perm(e[3], e[2], 0x0C0C0400u);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
  };
#pragma unroll 1
  for (int t = 0; t < 16 / PER; ++t) {
#pragma unroll
    for (int i = 0; i < PER; ++i) out[t * PER + i] = dword(4 * i);
  }
#pragma unroll
  for (int d = 16 / PER * PER; d < 16; ++d) out[d] = dword(4 * d);  // (K = 3: the sixteenth dword; 60 is a multiple of 3)
  used = GHF_WINDOW_USED();
  return acc;
}

// HOT, small alphabet (2 * max_len <= 10): any two codes fit lut2's index, so one lookup yields two symbols and the serial
// shift -> lookup -> add chain is half as long.  Three lookups (<= 30 bits) per refill check: a trip is six lookups, three dwords.
// entry = sym0 | sym1 << 8 | (len0 + len1) << 16 | bit 30: not two data symbols
__device__ __forceinline__ uint32_t dec_hot_pair(const uint8_t* lin, uint32_t la0, const DecLut& T2, uint32_t pos, DecOut& out, uint32_t& used) {
  GHF_WINDOW_OPEN();
  uint32_t acc = 0;
  auto dword = [&](int look0) -> uint32_t {  // lookups look0, look0 + 1 of the trip -> four symbols
    uint32_t e[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if ((look0 + j) % 3 == 0) GHF_REFILL();
      const uint32_t ent = dec_lookup(T2, (uint32_t)((W << o) >> 32));
      o += (ent >> 16) & 0xFFu;
      e[j] = ent;
    }
    acc |= e[0] | e[1];
    return __builtin_amdgcn_perm(e[1], e[0], 0x05040100u);  // {a.sym0, a.sym1, b.sym0, b.sym1}
  };
#pragma unroll 1
  for (int t = 0; t < 5; ++t) {
#pragma unroll
    for (int i = 0; i < 3; ++i) out[t * 3 + i] = dword(2 * i);
  }
  out[15] = dword(30);
  used = GHF_WINDOW_USED();
  return acc;
}

#undef GHF_REFILL
#undef GHF_WINDOW_OPEN
#undef GHF_WINDOW_USED

// COLD: whatever the hot passes do not take -- the stream's last group (ragged, followed by the end mark), spans that do
// not fit the LDS tile (read from memory), unaligned output.  One symbol at a time, one byte per store.
template <bool STAGED>
__device__ __forceinline__ uint32_t dec_cold(const DecLds7& L, const DecIn<STAGED>& I, const DecLut& T, int lut_bits, int max_len,
                                             uint64_t pos, uint32_t cnt, bool valid, uint8_t* optr, int has_next,
                                             uint64_t expect_bits) {
  if (!valid) return 0u;
  uint32_t widx = (uint32_t)(pos >> 5);
  uint32_t o = (uint32_t)(pos & 31u);
  const uint32_t o0 = o, widx0 = widx;
  uint64_t W = ((uint64_t)I.fetch(widx) << 32) | I.fetch(widx + 1);
  uint32_t nextw = I.fetch(widx + 2);
  widx += 3;
  uint32_t acc = 0;
  auto one = [&]() -> uint32_t {
    while (o >= 32u) {  // (a code of up to 64 bits moves the cursor by up to two words)
      W = (W << 32) | nextw;
      o -= 32u;
      nextw = I.fetch(widx++);
    }
    const uint32_t v = (uint32_t)((W << o) >> 32);
    uint32_t ent = dec_lookup(T, v);
    if (ent & kEntNone) ent = dec_long_entry_at(L, W, nextw, o, lut_bits, max_len);
    o += (ent >> 8) & 0xFFu;
    return ent;
  };
  for (uint32_t i = 0; i < cnt; ++i) {
    const uint32_t ent = one();
    acc |= ent;
    optr[i] = (uint8_t)ent;
  }
  // the index says where the next segment starts: an end-to-end check of every segment
  if (has_next == 1) {
    const uint64_t used = (uint64_t)(widx - widx0 - 3) * 32 + o - o0;
    if (used != expect_bits) acc |= kEntNone;
  } else if (has_next == 0) {
    if (!(one() & kEntEnd)) acc |= kEntNone;  // canonical_huff_encoder.cc:404: the end mark must follow
  }
  return acc;
}

// K7.  Persistent waves; each pass a wave takes one side-car block = 64 consecutive segments (4096 symbols):
//   1. the compressed span of the block (known from the side-car) is copied into LDS with coalesced 16-byte loads,
//      byte-swapped to big-endian words;
//   2. every lane decodes its 64 symbols from a 64-bit window: one LDS table lookup per symbol, the 64 bytes stay in
//      registers;
//   3. the wave's 4 KiB of output go through the (now dead) input tile and leave as four coalesced 1 KiB stores.
// The loop is software-pipelined over groups so that no HBM latency is exposed and nothing but the copy into LDS stands
// between the arrival of a span and the request for the next one: while group i is decoded, the span of group i+1 is in
// flight into registers, the descriptor of group i+1 (where to load, where every lane starts) was computed a pass earlier,
// the side-car words of group i+2 are in flight and the ticket for group i+3 is in flight.
//
// Round 3's form of this loop computed the next group's descriptor between the copy into LDS and the loads, held six
// instantiations of the whole loop nest (one per decoder variant, each with its own cold path) and was spilled by the
// register allocator INSIDE the hot loop of the variants uniform bytes take: the fifth vector of the prefetch went to
// scratch right behind its load, i.e. behind an s_waitcnt vmcnt(0) -- every pass waited for the whole prefetch before it
// decoded a symbol, the loads never overlapped the decode (profiles/r04/k7_spill_r03.txt has the listing).  Now: ONE loop,
// the variant is a switch around the 64 lookups only, one cold path, and a CPU-side guard keeps the kernel scratch-free
// (tests/test_cabi_cpu.py).
struct DecMeta {   // side-car words of one group, as loaded: issued a whole pass before they are combined
  uint64_t blk;    // block start (same word in every lane)
  uint32_t end;    // where MY segment ends, relative to blk
};

constexpr uint32_t kDecBadSeg = 0xFFFFFFFFu;
struct DecGroup {       // one group, ready to be fetched and decoded
  const uint8_t* base;  // uniform: what the span loads are relative to (the span's first byte, 16-aligned)
  uint32_t lim;         // uniform: the vector at byte offset o of the span is loaded iff o + 16 <= lim
  uint64_t byte0;       // uniform: first staged byte (16-aligned)
  uint32_t span;        // uniform: staged bytes
  uint32_t pos;         // bit of the staged span at which my segment starts
  uint32_t expect;      // bits of my segment; kDecBadSeg: my side-car words are implausible (no decode consumes that many bits)
  bool hot;             // uniform: complete, staged, plausible, not the stream's last group
};

struct DecConst {  // wave-uniform facts of one launch
  uint64_t n_segs, stream_bytes, stream_end_bit, full_bytes;
  uint32_t ngroups;
  int max_len;
  bool hot_ok;  // the output is 16-byte aligned and no code is longer than 32 bits
};

__device__ __forceinline__ void dec_issue_meta(const DecParams& P, const DecConst& C, uint32_t group, int lane, DecMeta& M) {
  const uint64_t last = C.n_segs - 1;
  const uint64_t seg = (uint64_t)group * 64 + lane;
  M.blk = P.chunk_bit[group];
  M.end = P.seg_bit[seg < last ? seg : last];  // clamped: unconditional loads
}

__device__ __forceinline__ void dec_group(const DecParams& P, const DecConst& C, uint32_t group, int lane, const DecMeta& M, DecGroup& G) {
  const uint64_t seg0 = (uint64_t)group * 64;
  const bool valid = seg0 + lane < C.n_segs;
  const uint64_t B0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(M.blk >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)M.blk);
  // my segment starts where my left neighbour's ends (v_mov_dpp wave_shr:1; lane 0 starts with the block)
  const uint32_t start = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)M.end, 0x138, 0xF, 0xF, true);
  uint64_t B1 = B0 + (uint32_t)__builtin_amdgcn_readlane((int)M.end, 63);
  const bool last_group = seg0 + 64 >= C.n_segs;
  if (last_group) {
    // last group: bound it by its last segment's worst case (+ end mark) rather than by a side-car word.  Segment starts
    // grow with the lane (a side-car that says otherwise is caught by `bad`), so the last valid lane's start is the largest
    const uint32_t lastv = (uint32_t)(C.n_segs - 1 - seg0) & 63u;
    B1 = B0 + (uint32_t)__builtin_amdgcn_readlane((int)start, (int)lastv) + 65u * (uint32_t)C.max_len;
  }
  if (B1 > C.stream_end_bit) B1 = C.stream_end_bit;
  G.byte0 = (B0 >> 3) & ~15ull;
  uint64_t byte1 = ((B1 + 7) >> 3) + 12;  // window look-ahead
  if (byte1 > C.stream_bytes) byte1 = C.stream_bytes;
  if (G.byte0 > byte1) G.byte0 = byte1 & ~15ull;  // corrupt side-car: caught by `bad`
  const uint64_t span = byte1 - G.byte0;
  G.span = span > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)span;
  G.pos = (uint32_t)(B0 & 127u) + start;
  const bool bad = valid && (M.end < start || B0 + M.end > C.stream_end_bit || B0 >= C.stream_end_bit);
  G.expect = bad ? kDecBadSeg : M.end - start;
  // vector k of a lane = bytes byte0 + k * 1024 + lane * 16 .. of the stream; it is loaded when it begins inside the span
  // and ends inside the stream's whole 16-byte vectors:  o + 16 <= lim  <=>  o < span && byte0 + o + 16 <= full_bytes
  uint32_t lim = 0;
  if (G.byte0 <= C.full_bytes) {
    const uint64_t room = C.full_bytes - G.byte0;
    const uint64_t a = (uint64_t)G.span + 15u;
    lim = (uint32_t)(a < room ? a : (room > 0x7FFFFFFFull ? 0x7FFFFFFFull : room));
  }
  G.lim = lim;
  G.base = P.stream + (lim ? G.byte0 : 0ull);
  G.hot = C.hot_ok && !last_group && G.span <= (uint32_t)kDec7InBytes && G.byte0 + G.span <= C.full_bytes && __ballot(bad) == 0;
}

// latch_status whose operands are materialised where it stands (the optimiser otherwise builds the compare-and-swap's
// register pair in front of the loop and keeps it alive -- in scratch -- across all of it)
__device__ __forceinline__ void latch_status_here(int* st, int code) {
  int want = 0;
  asm volatile("" : "+v"(code), "+v"(want));
  atomicCAS(st, want, code);
}

constexpr int kDecVec = (kDec7InBytes + 1023) / 1024;  // 16-byte vectors per lane that cover a staged span

__global__ __launch_bounds__(kDec7Threads) void k_decode(DecParams P) {
  __shared__ DecLds7 L;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform -> scalar loop control
  DecConst C;
  C.n_segs = P.n_segs;
  C.stream_bytes = P.stream_bytes;
  C.stream_end_bit = P.stream_bytes * 8;
  C.full_bytes = P.stream_bytes & ~15ull;  // whole 16-byte vectors of the stream
  // (group numbers are 32-bit -- launch_decode refuses more -- so that the loop control is scalar compares of single
  //  registers: gfx9 has no scalar 64-bit ordering compare)
  C.ngroups = (uint32_t)((P.n_segs + 63) >> 6);
  const uint32_t ngroups = C.ngroups, glast = ngroups - 1;
  auto clampg = [&](uint32_t g) { return g < ngroups ? g : glast; };  // past the end: redundant, harmless loads
  // Groups.  A wave owns ONE group by its number (wid); every other one comes from a ticket counter, so that a workgroup that
  // starts late (a 16-wave workgroup needs a whole CU: one wave of another kernel on it -- K2 of a later step in the
  // pipelined bench -- and it waits for that kernel or for another K7 workgroup to end) owns 16 groups, not more: round 3 gave
  // every wave three groups and the first form of this loop four, and in the pipelined bench that workgroup's 64 groups ran
  // alone behind everybody else's last one (decode 0.135 ms alone, 0.159 in the pipeline).  16 classes of workgroups, one
  // counter each on its own 128-byte line (one word saturates at ~88 tickets per microsecond); class c owns the ticketed
  // groups == c (mod 16).  The first ticket is worth two groups (the depth of the pipeline) and is drawn before the
  // tables are copied in, which hides its round trip.
  const uint32_t nwaves = gridDim.x * kDec7Waves;
  const uint32_t wid = blockIdx.x * kDec7Waves + (uint32_t)wave;
  // A class's workgroups are spread over all eight XCDs (consecutive workgroups go to consecutive XCDs): a class is a
  // fixed share of the groups, and an XCD that runs slower than the others would otherwise finish its classes last.
  const uint32_t ncls = gridDim.x < 16u ? 1u : (gridDim.x < 128u ? gridDim.x >> 3 : 16u);  // every class needs at least one workgroup
  const uint32_t cls = (blockIdx.x >> 3) % ncls;
  uint32_t* const my_ticket = &P.dt->ticket[cls * 32];
  auto claim_issue = [&](unsigned int count) -> unsigned int {  // the atomic's return value stays in a VGPR until claim_group() needs it, a pass later
    unsigned int t = 0;
    if (lane == 0) t = atomicAdd(my_ticket, count);
    return t;
  };
  auto claim_group = [&](unsigned int t, uint32_t k) -> uint32_t {  // group of ticket t + k (saturating: past the end stays past the end)
    const uint64_t g = (uint64_t)nwaves + ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)t) + k) * ncls + cls;
    return g < 0xFFFFFFFFull ? (uint32_t)g : 0xFFFFFFFFu;
  };
  uint32_t g0 = wid, g1, g2;
  const unsigned int tk0 = claim_issue(2);

  // The first two groups' side-car words are requested BEFORE the tables are pulled in: one of the dependent memory round
  // trips in front of the first decode hides behind the table copy.
  DecMeta M0, M;
  dec_issue_meta(P, C, clampg(g0), lane, M0);
  if (tid == 0) L.status0 = *P.status;  // one read per workgroup: whether the launch does anything must be uniform
  const int lut_bits = P.dt->lut_bits, max_len = P.dt->max_len;
  const int pair_bits = P.dt->pair_bits;
  dec_lds_load7(L, P.dt, tid, kDec7Threads);
  if (blockIdx.x == 0 && tid == 0 && P.out_bytes) *P.out_bytes = P.n_symbols;
  __syncthreads();
  C.max_len = max_len;
  C.hot_ok = (((uintptr_t)P.out) & 15u) == 0 && max_len <= 32;  // codes beyond 32 bits (a very deep .crs tree): one symbol at a time

  if (L.status0 == 0 && g0 < ngroups) {
    // a lane's replicas of the tables: T1 = one symbol per lookup, T2 = two (small alphabets only).  The uniform parts live in
    // SGPRs, the lane's part is recomputed in every pass (two VALU instructions instead of two registers)
    const int r1 = pair_bits ? 5 : ((kDec7LutLog2 - lut_bits) < 5 ? (kDec7LutLog2 - lut_bits) : 5);
    const int r2 = (kDec7LutLog2 - pair_bits) < 5 ? (kDec7LutLog2 - pair_bits) : 5;
    const uint32_t* const t1 = L.lut + (pair_bits ? kDec7LutSlots : 0);
    auto lut1 = [&](int ln) {
      DecLut T;
      T.base = reinterpret_cast<const char*>(t1 + ((uint32_t)ln & ((1u << r1) - 1u)));
      T.lsh = 32 - lut_bits;
      T.ash = r1 + 2;
      return T;
    };
    auto lut2 = [&](int ln) {
      DecLut T;
      T.base = reinterpret_cast<const char*>(L.lut + ((uint32_t)ln & ((1u << r2) - 1u)));
      T.lsh = 32 - pair_bits;
      T.ash = r2 + 2;
      return T;
    };
    // decoder variant (wave-uniform): 0 pair table; 1..3 K = 4 / 3 / 2 symbols per refill check; 4, 5 codes beyond the table
    // K = 32 / max_len symbols per refill check (o <= 31 behind a check, o + K * max_len <= 63 before the next: a single
    // refill brings it back below 32.  With 33 -- max_len 11, K = 3 -- o could reach 64, stay at 32 behind the refill, and
    // the third lookup of the next round would read past the window: six 11-bit codes in a row at the right phase, found by
    // scratch/host_soak.py)
    const int var = pair_bits ? 0 : max_len <= 8 ? 1 : max_len <= 10 ? 2 : max_len <= kDecLutBitsMax ? 3 : max_len <= 16 ? 4 : 5;
    const uint8_t* const lin = L.in;
    const uint32_t la0 = (uint32_t)wave * kDec7TileLog;             // this wave's tile in the logical (unpadded) byte space
    uint32_t* const tile = reinterpret_cast<uint32_t*>(L.in + (uint32_t)wave * kDec7TilePhys);  // ... and as plain memory (copy-out)
    uint32_t bad_acc = 0;

    auto issue = [&](const DecGroup& G, uint4 (&R)[kDecVec], int ln) {
#pragma unroll
      for (int k = 0; k < kDecVec; ++k) {  // lanes behind the span re-read the span's first bytes (an L2 hit)
        const uint32_t o = (uint32_t)k * 1024u + (uint32_t)ln * 16u;
        R[k] = load_stream(G.base + (o + 16u <= G.lim ? o : 0u));  // read once: not worth a line of the Infinity Cache
      }
    };

    DecGroup cur, nxt;
    uint4 R[kDecVec];
    dec_group(P, C, clampg(g0), lane, M0, cur);
    issue(cur, R, lane);
    g1 = claim_group(tk0, 0);
    g2 = claim_group(tk0, 1);
    dec_issue_meta(P, C, clampg(g1), lane, M);
    unsigned int tk = claim_issue(1);  // for the group after g2
    dec_group(P, C, clampg(g1), lane, M, nxt);  // (the one dependent side-car round trip of a wave's life; the first span is in flight meanwhile)
    dec_issue_meta(P, C, clampg(g2), lane, M);

    // One pass over a group.  HOT = the group is complete, staged, plausible and not the stream's last: the body then has
    // no data-dependent branch around its memory operations, so the compiler can count them -- the wait for the
    // prefetched span becomes "all but the youngest seven" (this group's output stores, the next side-car words, the
    // ticket) instead of vmcnt(0), and the wave never sleeps until its own stores are acknowledged by L2.
    auto pass = [&](auto hot_tag) {
      constexpr bool HOT = decltype(hot_tag)::value;
      // the lane number, recomputed (v_mbcnt) and opaque to the optimiser: everything derived from it below (a dozen LDS and
      // global addresses) costs a few VALU instructions per pass instead of registers that live across the whole loop
      int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      asm volatile("" : "+v"(ln));
      // ---- 1. this group's span: registers -> LDS (big-endian words).  HOT: whatever a lane loaded behind the span is
      // harmless (only a corrupt stream reads it, and that is caught by the segment-end check); else it reads as zero
      wave_sync();
#pragma unroll
      for (int k = 0; k < kDecVec; ++k) {
        const uint32_t o = (uint32_t)k * 1024u + (uint32_t)ln * 16u;
        uint4 v = R[k];
        v = make_uint4(bswap32(v.x), bswap32(v.y), bswap32(v.z), bswap32(v.w));
        if (!HOT && !(o + 16u <= cur.lim)) v = make_uint4(0, 0, 0, 0);
        if ((k + 1) * 1024 <= kDec7InBytes || o < (uint32_t)kDec7InBytes)
          *reinterpret_cast<uint4*>(L.in + in_phys(la0 + o)) = v;
      }
      if (ln < 4) *reinterpret_cast<uint32_t*>(L.in + in_phys(la0 + kDec7InBytes + 4u * ln)) = 0;
      if (!HOT && cur.byte0 + cur.span > C.full_bytes && C.full_bytes >= cur.byte0 && ln == 0) {
        // the stream's last, incomplete 16 bytes: byte loads, never past the end of the buffer
        uint32_t q[4] = {0, 0, 0, 0};
        for (uint64_t j = 0; C.full_bytes + j < C.stream_bytes; ++j) q[j >> 2] |= (uint32_t)P.stream[C.full_bytes + j] << (24 - 8 * (j & 3));
        const uint64_t w = (C.full_bytes - cur.byte0) >> 2;
        if (w + 3 < (uint64_t)kDec7InWords + 4) {
          for (int j = 0; j < 4; ++j) *reinterpret_cast<uint32_t*>(L.in + in_phys(la0 + 4u * (uint32_t)(w + j))) = q[j];
        }
      }
      wave_sync();
      // ---- 2. the next group's span is requested at once: its descriptor was computed a pass ago
      issue(nxt, R, ln);
      // ---- 3. decode
      const uint64_t seg0 = (uint64_t)g0 * 64;
      const uint64_t seg = seg0 + ln;
      const uint64_t sym0 = seg * kSegSymbols;
      if (HOT) {
        uint32_t used, acc;
        DecOut out;
        if (var == 0) acc = dec_hot_pair(lin, la0, lut2(ln), cur.pos, out, used) >> 14;  // bit 30 -> bit 16
        else if (var == 1) acc = dec_hot<4, false>(L, lin, la0, lut1(ln), lut_bits, max_len, cur.pos, out, used);
        else if (var == 2) acc = dec_hot<3, false>(L, lin, la0, lut1(ln), lut_bits, max_len, cur.pos, out, used);
        else if (var == 3) acc = dec_hot<2, false>(L, lin, la0, lut1(ln), lut_bits, max_len, cur.pos, out, used);
        else if (var == 4) acc = dec_hot<2, true>(L, lin, la0, lut1(ln), lut_bits, max_len, cur.pos, out, used);
        else acc = dec_hot<1, true>(L, lin, la0, lut1(ln), lut_bits, max_len, cur.pos, out, used);
        // copy-out through the input tile (dead now): lane-major 64-byte rows, pieces XOR-swizzled so that the 16
        // lanes of a write phase hit 16 different bank groups; then four fully coalesced 1 KiB stores per wave,
        // straight-line, so that the compiler can count them
        wave_sync();
        const uint32_t osw = ((uint32_t)ln >> 2) & 3u;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<uint4*>(tile + ln * 16 + (((uint32_t)q ^ osw) << 2)) = make_uint4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
        wave_sync();
        uint8_t* og = P.out + seg0 * kSegSymbols + (uint32_t)ln * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t sl = (uint32_t)r * 16 + ((uint32_t)ln >> 2);  // the lane whose row holds my piece
          const uint32_t piece = ((uint32_t)ln & 3u) ^ ((sl >> 2) & 3u);
          store_stream(og + r * 1024, *reinterpret_cast<const uint4*>(tile + sl * 16 + piece * 4));
        }
        if (used != cur.expect) acc |= kEntNone;
        bad_acc |= acc;
      } else {
        const bool valid = seg < C.n_segs;
        if (__ballot(cur.expect == kDecBadSeg)) {
          if (cur.expect == kDecBadSeg) latch_status_here(P.status, GHF_E_CORRUPT);
        } else {
          const bool staged = cur.span <= (uint32_t)kDec7InBytes;
          uint32_t cnt = 0;
          if (valid) cnt = (P.n_symbols - sym0 >= (uint64_t)kSegSymbols) ? (uint32_t)kSegSymbols : (uint32_t)(P.n_symbols - sym0);
          // 1: the side-car says where the next segment starts; 0: the end mark must follow; 2: nothing to check
          const int has_next = seg + 1 < C.n_segs ? 1 : (P.no_end_mark ? 2 : 0);
          const uint8_t* src = P.stream + cur.byte0;
          if (staged) {
            DecIn<true> I{lin, la0, src, cur.span};
            bad_acc |= dec_cold<true>(L, I, lut1(ln), lut_bits, max_len, cur.pos, cnt, valid, P.out + sym0, has_next, cur.expect);
          } else {
            DecIn<false> I{lin, la0, src, cur.span};
            bad_acc |= dec_cold<false>(L, I, lut1(ln), lut_bits, max_len, cur.pos, cnt, valid, P.out + sym0, has_next, cur.expect);
          }
        }
      }
      // ---- 4. behind the decode, where nothing waits for it: the descriptor of the group after the next (its side-car
      // words were requested a pass ago), the side-car words of the one after that, the number of the one after that
      cur = nxt;
      dec_group(P, C, clampg(g2), ln, M, nxt);
      g0 = g1;
      g1 = g2;
      g2 = claim_group(tk, 0);  // (drawn a pass ago: a wave holds two groups beyond the one it decodes -- what it still has to do
                                //  when the counters run dry is the launch's tail)
      dec_issue_meta(P, C, clampg(g2), ln, M);
      tk = claim_issue(1);
    };
    while (g0 < ngroups) {
      if (cur.hot) {
        // drain once on entry: the hot loop's waits are then computed from its own back edge alone (exact counts)
        // instead of being merged with whatever the cold paths left outstanding
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        do pass(std::true_type{});
        while (g0 < ngroups && cur.hot);
      }
      if (g0 < ngroups) pass(std::false_type{});
    }
    if (bad_acc & (kEntEnd | kEntNone)) latch_status_here(P.status, GHF_E_CORRUPT);  // 64 data symbols per full segment, always
  }
  // the last workgroup to finish hands the ticket counters back as it found them
  __syncthreads();
  if (tid == 0) {
    const unsigned int arrived = atomicAdd(&P.dt->done, 1u);
    if (arrived == gridDim.x - 1) {
      for (int k = 0; k < 16; ++k) __hip_atomic_store(&P.dt->ticket[k * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&P.dt->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

void launch_decode(const DecParams& p, hipStream_t s) {
  const uint64_t groups = (p.n_segs + 63) / 64;
  uint64_t blocks = (groups + kDec7Waves - 1) / kDec7Waves;
  if (blocks == 0) return;
  if (groups >= kDecMaxGroups) return;  // (k_decode numbers its groups in 32 bits; the callers refuse such an index first)
  if (blocks > 256) blocks = 256;  // persistent: one workgroup of 16 waves per CU (its LDS tiles + table take 153 KiB)
  hipLaunchKernelGGL(k_decode, dim3((uint32_t)blocks), dim3(kDec7Threads), 0, s, p);
}

// ------------------------------------------------------------------------------------------------
// K6: rebuild the side-car of a FOREIGN stream (a .crs2 written by the reference has no sync points).
// Huffman codes self-synchronise: a decoder started at a wrong bit falls into step with the true
// code boundaries after a few symbols.  The body is cut into 512-bit subsequences; every thread decodes
// its subsequence from its current guess of the first code boundary and tells its right neighbour where
// it landed.  Thread 0 starts at a true boundary, so the fixed point of this iteration is the true
// segmentation; passes repeat (only threads whose guess changed redo work) until nothing changes.
// Then symbol counts are prefix-summed, the end mark fixes n, and one more pass writes the bit position
// of every 64th symbol -- the same side-car K5 emits.
// All K6 kernels run on K7's engine: one 16-wave workgroup per CU, the 64 KiB replicated direct table
// (conflict-free lookups), padded input tiles (conflict-free window refills); a wave trip = 64
// subsequences = 4 KiB of stream.
// ------------------------------------------------------------------------------------------------
constexpr int kSubBits = 512;

// this lane's replica of the one-symbol table (K7's T1)
__device__ __forceinline__ DecLut dec7_lut1(const DecLds7& L, const DecTables* dt, int lane) {
  const int pair_bits = dt->pair_bits, lut_bits = dt->lut_bits;
  const int r1 = pair_bits ? 5 : ((kDec7LutLog2 - lut_bits) < 5 ? (kDec7LutLog2 - lut_bits) : 5);
  const uint32_t* t1 = L.lut + (pair_bits ? kDec7LutSlots : 0);
  DecLut T;
  T.base = reinterpret_cast<const char*>(t1 + ((uint32_t)lane & ((1u << r1) - 1u)));
  T.lsh = 32 - lut_bits;
  T.ash = r1 + 2;
  return T;
}

// a lane's decode cursor over its wave's staged tile: 64-bit window + one word of look-ahead
struct K6Cursor {
  uint32_t la;  // logical byte address of the next word to fetch
  uint64_t W;
  uint32_t nextw, o;
  __device__ __forceinline__ void open(const uint8_t* lin, uint32_t la0, uint32_t pos) {
    la = la0 + ((pos >> 5) << 2);
    o = pos & 31u;
    W = ((uint64_t)in_word(lin, la) << 32) | in_word(lin, la + 4u);
    nextw = in_word(lin, la + 8u);
    la += 12u;
  }
  // the entry (symbol | length << 8 | flags) of the code at the cursor; the cursor moves behind it.
  // lut_bits < 0: the caller knows that the direct table resolves every code (max_len <= 12, canonical): no miss path
  __device__ __forceinline__ uint32_t step(const uint8_t* lin, const DecLds7& L, const DecLut& T, int lut_bits, int max_len) {
    if (o >= 32u) {
      W = (W << 32) | nextw;
      o -= 32u;
      nextw = in_word(lin, la);
      la += 4u;
    }
    if (lut_bits >= 0 && o >= 32u) {  // (only a code beyond 32 bits moves the cursor by two words)
      W = (W << 32) | nextw;
      o -= 32u;
      nextw = in_word(lin, la);
      la += 4u;
    }
    const uint32_t v = (uint32_t)((W << o) >> 32);
    uint32_t ent = dec_lookup(T, v);
    if (lut_bits >= 0 && (ent & kEntNone)) ent = dec_long_entry_at(L, W, nextw, o, lut_bits, max_len);
    o += (ent >> 8) & 0xFFu;
    return ent;
  }
  // the same for a canonical code with at most TWO lengths (uniform bytes: 8 / 9 bits; 16 symbols: 4 / 5): which of the two a
  // code has is one comparison of the next bits with the first code of the shorter length -- no table, no LDS round trip
  // in the dependency chain (canonical_huff_encoder.cc:446-450: the first length whose left-justified first code is <= v).
  // Returns length << 8 (| kEntEnd for the end mark): all that K6 ever asks of an entry.
  __device__ __forceinline__ uint32_t step2(const uint8_t* lin, const struct K6Two& C);
};
struct K6Two {
  uint32_t thr;      // first code of the shorter length, left-justified (0: one length only)
  uint32_t lmin;     // the shorter length
  uint32_t eof_lo;   // the end mark's code, left-justified ...
  uint32_t eof_span; // ... and 2^(32 - its length): v - eof_lo < eof_span  <=>  the next code IS the end mark
};
__device__ __forceinline__ uint32_t K6Cursor::step2(const uint8_t* lin, const K6Two& C) {
  if (o >= 32u) {
    W = (W << 32) | nextw;
    o -= 32u;
    nextw = in_word(lin, la);
    la += 4u;
  }
  const uint32_t v = (uint32_t)((W << o) >> 32);
  const uint32_t len = C.lmin + (v < C.thr ? 1u : 0u);
  o += len;
  return (len << 8) | ((v - C.eof_lo < C.eof_span) ? kEntEnd : 0u);
}
// MODE 0: table + tree walk / linear extension, 1: the table resolves every code, 2: two lengths
template <int MODE>
__device__ __forceinline__ uint32_t k6_step(K6Cursor& c, const uint8_t* lin, const DecLds7& L, const DecLut& T, int lut_bits, int max_len,
                                            const K6Two& C2) {
  if (MODE == 2) return c.step2(lin, C2);
  return c.step(lin, L, T, MODE == 1 ? -1 : lut_bits, max_len);
}

// Two codes of at most 16 bits behind ONE refill check (table + miss path, or table only): their entries; the window may
// move, the position (o) does not -- the caller commits one or both.  Half the refill / bounds / end-mark tests of two
// single steps: the generic K6 loops run at 2.3x K7's time per decoded symbol, and most of that is tests, not lookups.
template <int MODE>
__device__ __forceinline__ void k6_peek2(K6Cursor& c, const uint8_t* lin, const DecLds7& L, const DecLut& T, int lut_bits, int max_len,
                                         uint32_t& e0, uint32_t& e1) {
  static_assert(MODE == 0 || MODE == 1, "two-length codes step by comparison");
  if (c.o >= 32u) {
    c.W = (c.W << 32) | c.nextw;
    c.o -= 32u;
    c.nextw = in_word(lin, c.la);
    c.la += 4u;
  }
  const uint32_t v0 = (uint32_t)((c.W << c.o) >> 32);
  e0 = dec_lookup(T, v0);
  if (MODE == 0 && (e0 & kEntNone)) e0 = dec_long_entry(L, v0, lut_bits, max_len);
  const uint32_t v1 = (uint32_t)((c.W << (c.o + ((e0 >> 8) & 0xFFu))) >> 32);  // (o < 32, a code <= 16 bits: >= 16 valid bits)
  e1 = dec_lookup(T, v1);
  if (MODE == 0 && (e1 & kEntNone)) e1 = dec_long_entry(L, v1, lut_bits, max_len);
}

// stage the bits of 64 consecutive subsequences (+ look-ahead) of the body into the wave's padded tile (big-endian
// words, zeros behind the stream); returns the bit offset of subsequence `sub0` inside the tile
__device__ __forceinline__ uint32_t k6_stage(const SyncParams& P, uint64_t sub0, uint8_t* lin, uint32_t la0, int lane) {
  const uint64_t bit0 = P.body_bit0 + sub0 * kSubBits;
  const uint64_t byte0 = (bit0 >> 3) & ~15ull;
  uint64_t byte1 = ((bit0 + 64ull * kSubBits + 7) >> 3) + 32;  // look-ahead: a code of <= 64 bits that begins in the last subsequence + the cursor's three words
  if (byte1 > P.stream_bytes) byte1 = P.stream_bytes;
  const uint32_t span = byte1 > byte0 ? (uint32_t)(byte1 - byte0) : 0u;
  const uint8_t* src = P.stream + byte0;
#pragma unroll
  for (int k = 0; k < (kDec7TileLog + 1023) / 1024; ++k) {
    const uint32_t o = (uint32_t)k * 1024u + (uint32_t)lane * 16u;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (o + 16u <= span) {
      v = *reinterpret_cast<const uint4*>(src + o);
    } else if (o < span) {  // the stream's last, incomplete 16 bytes: byte loads, never past the end of the buffer
      uint32_t q[4] = {0, 0, 0, 0};
      for (uint32_t j = 0; o + j < span; ++j) q[j >> 2] |= (uint32_t)src[o + j] << (8 * (j & 3));
      v = make_uint4(q[0], q[1], q[2], q[3]);
    }
    if (o + 16u <= (uint32_t)kDec7TileLog)
      *reinterpret_cast<uint4*>(lin + in_phys(la0 + o)) = make_uint4(bswap32(v.x), bswap32(v.y), bswap32(v.z), bswap32(v.w));
  }
  return (uint32_t)(bit0 - byte0 * 8);
}

// the two-length facts of a canonical code (meaningless, and unused, for any other code): from the tables in LDS
__device__ __forceinline__ K6Two k6_two(const DecLds7& L, int min_len, int max_len) {
  K6Two C;
  C.lmin = (uint32_t)min_len;
  C.thr = max_len > min_len ? L.fcl[min_len & 31] : 0u;
  // the end mark is the largest symbol, hence the LAST code of its length: it sits at the end of its length's run in symbol[]
  int k = 0;
  while (k < GHF_NSYM && L.symbol[k] != 256) ++k;
  const int len = (max_len > min_len && k >= (int)L.sp[max_len & 31]) ? max_len : min_len;
  const uint32_t code_left = L.fcl[len & 31] + (((uint32_t)k - L.sp[len & 31]) << ((32 - len) & 31));
  C.eof_lo = k < GHF_NSYM ? code_left : 0xFFFFFFFFu;
  C.eof_span = k < GHF_NSYM ? (1u << ((32 - len) & 31)) : 0u;
  return C;
}

constexpr int kK6Threads = kDec7Threads;
constexpr int kK6Waves = kDec7Waves;

// what every K6 kernel starts with: tables into LDS, this lane's table replica, the wave's tile
#define GHF_K6_PROLOGUE()                                                  \
  __shared__ DecLds7 L;                                                    \
  const int tid = threadIdx.x;                                             \
  dec_lds_load7(L, P.dt, tid, kK6Threads);                                 \
  __syncthreads();                                                         \
  const int lane = tid & 63;                                               \
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);               \
  const int lut_bits = P.dt->lut_bits, max_len = P.dt->max_len;            \
  const bool direct = P.dt->kind == 0 && max_len <= kDecLutBitsMax;        \
  const int k6_mode = (P.dt->kind == 0 && max_len - P.dt->min_len <= 1 && max_len <= 31) ? 2 : (direct ? 1 : 0); \
  const K6Two C2 = k6_two(L, P.dt->min_len, max_len);                      \
  const DecLut T1 = dec7_lut1(L, P.dt, lane);                              \
  uint8_t* const lin = L.in;                                               \
  const uint32_t la0 = (uint32_t)wave * kDec7TileLog;                      \
  const uint64_t ngroups = (P.nsub + 63) >> 6;                             \
  const uint64_t body_bits = P.end_bit - P.body_bit0

__device__ __forceinline__ uint32_t k6_limit(uint64_t body_bits, uint64_t g) {  // end of the stream, relative to the wave's first subsequence
  const uint64_t l = body_bits - g * 64 * kSubBits;
  return l > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)l;
}

// ---- classes: K6 on canonical codes of L and L + 1 bits, L = 8 or 4 ---------------------------------------------------
// Bytes that do not compress -- 256 byte values + the end mark, none of them rare enough for a 10-bit code: 255 codes of 8
// bits and two of 9 (the least frequent value and the end mark: 9-bit codes 0 and 1, every 8-bit code >= 0x01) -- and the
// same shape one size down, 16 equally likely values + the end mark: 15 codes of 4 bits, two of 5.  A decoder that stands at
// bit p moves to p + L, or to p + L + 1 when the L bits at p are zero.  So it walks along the positions of its CLASS p mod 8
// (L = 4: of the two classes p mod 8 and p + 4 mod 8, alternately) until it meets L zero bits, and then along the next
// class: per subsequence a handful of jumps instead of 57 (128) steps -- and max_len chains of them in k_sync_table: these
// codes are the ones that re-synchronise slowest.  The 64 positions of class r are one 64-bit mask F[r] (bit 63 - i: the
// L bits at bit 8 i + r are zero), computed for all eight classes at once from the bit planes of the subsequence's 64 bytes
// (8 x 8 bit-matrix transposes: no step touches a single symbol); G[r] marks those of them whose next bit makes the code
// the end mark.
struct K6Cls {
  int L;             // 8 or 4: the code is of that kind (wave-uniform); 0: it is not
  uint32_t eof_bit;  // last bit of the end mark's code
};
__device__ __forceinline__ K6Cls k6_cls(int k6_mode, int min_len, int max_len, const K6Two& C2, uint64_t body_bit0) {
  K6Cls B;
  // (a body begins on a 4-byte boundary of a 16-byte aligned buffer: behind a .crs2 header, or at byte 0 of a piece)
  const bool shape = k6_mode == 2 && max_len == min_len + 1 && (min_len == 8 || min_len == 4) && C2.thr == (1u << (32 - min_len)) &&
                     C2.eof_span == (1u << (31 - min_len)) && C2.eof_lo < C2.thr && (body_bit0 & 31u) == 0;
  B.L = shape ? min_len : 0;
  B.eof_bit = (C2.eof_lo >> ((31 - min_len) & 31)) & 1u;
  return B;
}
__device__ __forceinline__ uint32_t bfi32(uint32_t m, uint32_t a, uint32_t b) { return (m & a) | (~m & b); }  // v_bfi_b32
// exchange the bits under m with the bits under m << d
__device__ __forceinline__ uint32_t delta_swap32(uint32_t x, uint32_t m, int d) { return bfi32(m, x >> d, bfi32(m << d, x << d, x)); }

// F (and G) of the subsequence that begins at bit `bitpos` (a multiple of 32: k6_cls) of the wave's staged tile
template <int L, bool WITH_EOF>
__device__ __forceinline__ void k6_classes(const uint8_t* lin, uint32_t la0, uint32_t bitpos, uint32_t eof_bit, uint64_t (&F)[8], uint64_t (&G)[8]) {
  static_assert(L == 8 || L == 4, "class length");
  const uint32_t la = la0 + ((bitpos >> 5) << 2);
  uint32_t E[17];  // the subsequence's 512 bits + 32 of look-ahead, first bit in bit 31 of E[0]
#pragma unroll
  for (int k = 0; k < 17; ++k) E[k] = in_word(lin, la + 4u * (uint32_t)k);
  // transpose every group of 8 bytes (rows = bytes, first byte on top): T[g][0] holds bit planes 0..3 (plane s = bit s of
  // every byte, most significant first; byte 3 - s of the word), T[g][1] planes 4..7
  uint32_t T[8][2];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    uint32_t hi = delta_swap32(delta_swap32(E[2 * g], 0x00AA00AAu, 7), 0x0000CCCCu, 14);
    uint32_t lo = delta_swap32(delta_swap32(E[2 * g + 1], 0x00AA00AAu, 7), 0x0000CCCCu, 14);
    T[g][0] = bfi32(0xF0F0F0F0u, hi, lo >> 4);
    T[g][1] = bfi32(0xF0F0F0F0u, hi << 4, lo);
  }
  // plane s of all 64 bytes: Q[s][0] = bytes 0..31, Q[s][1] = bytes 32..63 (byte i in bit 31 - i mod 32)
  constexpr int NQ = 8 + L + (WITH_EOF ? 0 : -1);  // planes 0 .. 7 + L (7 + L - 1 without the end-mark masks)
  uint32_t Q[16][2];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const uint32_t a = T[4 * hh][half], b = T[4 * hh + 1][half], c = T[4 * hh + 2][half], d = T[4 * hh + 3][half];
      const uint32_t abx = __builtin_amdgcn_perm(a, b, 0x07030602u), aby = __builtin_amdgcn_perm(a, b, 0x05010400u);
      const uint32_t cdx = __builtin_amdgcn_perm(c, d, 0x07030602u), cdy = __builtin_amdgcn_perm(c, d, 0x05010400u);
      Q[4 * half + 0][hh] = __builtin_amdgcn_perm(abx, cdx, 0x07060302u);
      Q[4 * half + 1][hh] = __builtin_amdgcn_perm(abx, cdx, 0x05040100u);
      Q[4 * half + 2][hh] = __builtin_amdgcn_perm(aby, cdy, 0x07060302u);
      Q[4 * half + 3][hh] = __builtin_amdgcn_perm(aby, cdy, 0x05040100u);
    }
  }
  // planes 8..: the same bits one byte later (bit s of bytes 1..64)
#pragma unroll
  for (int s = 0; s + 8 < NQ; ++s) {
    Q[s + 8][0] = alignbit(Q[s][0], Q[s][1], 31);
    Q[s + 8][1] = (Q[s][1] << 1) | ((E[16] >> (31 - s)) & 1u);
  }
  const uint32_t eofx = eof_bit ? 0u : 0xFFFFFFFFu;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    uint32_t f[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint32_t any = Q[r][h] | Q[r + 1][h] | Q[r + 2][h] | Q[r + 3][h];
      if (L == 8) any |= Q[r + 4][h] | Q[r + 5][h] | Q[r + 6][h] | Q[r + 7][h];
      f[h] = ~any;
    }
    F[r] = ((uint64_t)f[0] << 32) | f[1];
    if (WITH_EOF) G[r] = ((uint64_t)(f[0] & (Q[r + L][0] ^ eofx)) << 32) | (f[1] & (Q[r + L][1] ^ eofx));
  }
}
// where the masks of a wave's 64 subsequences live while its chains walk: F in the (unused) table area, [class][thread];
// G in the wave's own tile, whose bytes are in registers by then, [class][lane]
// (the masks are 64-bit words in arrays that are declared, and elsewhere accessed, as 32-bit words and bytes: may_alias
//  tells the compiler so -- without it type-based alias analysis may order these accesses freely against the others)
typedef uint64_t __attribute__((may_alias)) k6_mask_t;
__device__ __forceinline__ k6_mask_t* k6_f_slot(DecLds7& L, int tid) { return reinterpret_cast<k6_mask_t*>(L.lut) + tid; }
__device__ __forceinline__ k6_mask_t* k6_g_slot(DecLds7& L, int wave, int lane) {
  return reinterpret_cast<k6_mask_t*>(L.in + (uint32_t)wave * kDec7TilePhys) + lane;
}
static_assert(sizeof(DecLds7::lut) >= 8 * kDec7Threads * sizeof(uint64_t) && kDec7TilePhys >= 8 * 64 * (int)sizeof(uint64_t) && kDec7TilePhys % 8 == 0,
              "room for the class masks");
// One jump of a chain that stands at bit p < 512 of its subsequence: s = the L-bit codes in front of the next (L + 1)-bit
// code (found), or in front of the subsequence's end; eof = that code is the end mark.  fmask(c) / gmask(c) = the F / G mask
// of class c (from LDS; from registers when p is a compile-time constant).  All mask reads are issued together: a jump is
// ONE LDS round trip.
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

__global__ __launch_bounds__(kK6Threads, 4) void k_sync_pass(SyncParams P) {
  GHF_K6_PROLOGUE();
  auto run = [&](auto mode_tag) {
  constexpr int MODE = decltype(mode_tag)::value;
  for (uint64_t g = (uint64_t)blockIdx.x * kK6Waves + wave; g < ngroups; g += (uint64_t)gridDim.x * kK6Waves) {
    const uint64_t sub = g * 64 + lane;
    const bool valid = sub < P.nsub;
    uint32_t st = 0;
    bool work = false, moved = false;
    if (valid) {
      st = (P.first & 2u) ? (sub == 0 ? P.first_start : 0u) : P.start[sub];
      work = (P.first & 1u) ? true : P.used[sub] != st;
    }
    if (!__ballot(work)) continue;  // the whole wave's results are still current
    wave_sync();
    const uint32_t base = k6_stage(P, g * 64, lin, la0, lane);
    wave_sync();
    if (work) {
      const uint32_t sub_lo = (uint32_t)lane * kSubBits;  // relative to the wave's first subsequence
      const uint32_t sub_hi = sub_lo + kSubBits;
      const uint32_t limit = k6_limit(body_bits, g);
      const bool pairs = max_len <= 16 && limit >= 64u * kSubBits;  // (wave-uniform: short codes, every subsequence whole)
      uint32_t pos = sub_lo + st;
      uint32_t count = 0;
      bool eof = false;
      K6Cursor cur;
      cur.open(lin, la0, base + pos);
      if (MODE != 2 && pairs) {  // two codes per round while neither is an end mark; what is left goes one by one below
        while (pos < sub_hi) {
          uint32_t e0, e1;
          k6_peek2<MODE == 2 ? 1 : MODE>(cur, lin, L, T1, lut_bits, max_len, e0, e1);
          if ((e0 | e1) & kEntEnd) break;
          const uint32_t l0 = (e0 >> 8) & 0xFFu, l1 = (e1 >> 8) & 0xFFu;
          const bool both = pos + l0 < sub_hi;  // the second code begins in this subsequence too
          pos += both ? l0 + l1 : l0;
          cur.o += both ? l0 + l1 : l0;
          count += both ? 2u : 1u;
        }
      }
      while (pos < sub_hi && pos < limit) {
        const uint32_t ent = k6_step<MODE>(cur, lin, L, T1, lut_bits, max_len, C2);
        if (ent & kEntEnd) {  // the end mark (or bits that are no code)
          eof = true;
          break;
        }
        pos += (ent >> 8) & 0xFFu;
        ++count;
      }
      // .crs has no end mark: "eof" then means "this cannot be right" -- a bit pattern that is no code, or a last
      // code that runs past the end of the stream
      if (P.no_eof == 1u && pos > limit) eof = true;
      P.cnt[sub] = count;
      P.eof[sub] = eof ? 1 : 0;
      P.used[sub] = (uint16_t)st;
      if (P.first & 2u) {
        // the launch that fills `start`: every subsequence stores its neighbour's guess, 0 ("nothing known") included
        if (sub + 1 < P.nsub) {
          const uint16_t land = (!eof && pos >= sub_hi) ? (uint16_t)(pos - sub_hi) : (uint16_t)0;
          P.start[sub + 1] = land;
          if (land) {
            *P.changed = 1;
            moved = true;
          }
        } else if (P.no_eof != 2u) {
          P.start[P.nsub] = 0;  // (the landing slot: a piece's last subsequence stores it below)
        }
      } else if (!eof && pos >= sub_hi && sub + 1 < P.nsub) {
        const uint16_t land = (uint16_t)(pos - sub_hi);
        if (P.start[sub + 1] != land) {
          P.start[sub + 1] = land;
          *P.changed = 1;
          moved = true;
        }
      }
      // a PIECE of a stream (mode 2, multi-GPU decode): the last code may run into the next piece's bytes (they are
      // there as look-ahead); where it ends is the next piece's first code boundary
      if (P.no_eof == 2u && sub + 1 == P.nsub) P.start[P.nsub] = eof ? (uint16_t)0xFFFF : (uint16_t)(pos - limit);
    }
    // ... and roughly how MANY boundaries moved, for the driver's choice between more passes and the deterministic scan:
    // every 256th group adds its count to the word behind the flag (all groups would be three million atomics on one
    // address in a 4 GiB stream's first pass)
    const uint64_t mv = __ballot(moved);
    if (mv && lane == 0) {
      if ((g & 255u) == 0) atomicAdd(P.changed + 1, (uint32_t)__builtin_popcountll(mv));
      // ... and WHERE the first of them is: nothing in front of the stream's end mark moving any more is all the driver
      // needs (a buffer may go on behind its end mark -- stale bytes -- and those never have to settle).  Stored inverted
      // so that "none" is the zero the flag's memset leaves; the read in front keeps the atomics to the few that improve it
      const unsigned long long inv = ~(g * 64 + (uint64_t)__builtin_ctzll(mv));
      if (inv > __hip_atomic_load(P.moved_first_inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(P.moved_first_inv, inv);
    }
  }
  };
  if (k6_mode == 2) run(std::integral_constant<int, 2>{});
  else if (k6_mode == 1) run(std::integral_constant<int, 1>{});
  else run(std::integral_constant<int, 0>{});
}

// ---- K6 for streams that do not self-synchronise quickly (near-fixed-length codes: uniform bytes have 8/9-bit codes and a
// decoder started at a wrong bit needs ~1000 symbols to fall into step, so the fixed-point passes above advance a few
// subsequences per launch).  Deterministic instead: for every subsequence the landing offset of EVERY possible start
// offset (a code straddles a boundary by less than max_len <= 32 bits) is one small function; where the true decode
// enters each subsequence is the running composition of those functions -- a parallel scan over function composition
// (64-ary tree: reduce up, apply down).  The result only seeds start[]: the fixed-point passes then verify it in one
// pass (and would repair it), so no format subtlety (end mark, end of a .crs, stream pieces) lives here.
// A function is STRIDE bytes (16 when max_len <= 16, 32 up to 32, else 64): entry s = landing offset in the next subsequence.
//
// k_sync_table: lane = subsequence, its max_len chains five at a time -- independent shift -> lookup -> add
// dependency chains per lane hide the LDS round trip that a single chain leaves exposed (four waves per SIMD do not).
//
// Classes (k6_classes, codes of L and L + 1 bits): the max_len chains of a subsequence are walks over the class masks -- per
// chain and round one or two LDS reads of the masks of its class, a shift, a count of leading zeros.  The row then also
// says, for every entry offset, how many (L + 1)-bit codes the chain met and whether one of them was the end mark: with the
// entry offset that the scan arrives at, that IS the subsequence's symbol count -- the lowest k_fn_apply settles every
// subsequence whose true chain met no end mark, and the confirming pass has only the others left.
//   row, L = 8: bytes 0..8 landing offsets | 10..11 end-mark bits | 12..15 nine 3-bit counts (7 = seven or more)
//   row, L = 4: bytes 0..4 landing offsets | 5..9 five 8-bit counts (255 = or more) | 10..11 end-mark bits
__global__ __launch_bounds__(kK6Threads, 4) void k_sync_table(SyncParams P, uint32_t stride, uint8_t* __restrict__ tab, uint32_t* __restrict__ cls_flag) {
  GHF_K6_PROLOGUE();
  const uint32_t S = (uint32_t)max_len < stride ? (uint32_t)max_len : stride;
  const K6Cls CL = k6_cls(k6_mode, P.dt->min_len, max_len, C2, P.body_bit0);
  if (blockIdx.x == 0 && tid == 0) *cls_flag = (uint32_t)CL.L;
  auto run = [&](auto mode_tag) {
  constexpr int MODE = decltype(mode_tag)::value;
  for (uint64_t g = (uint64_t)blockIdx.x * kK6Waves + wave; g < ngroups; g += (uint64_t)gridDim.x * kK6Waves) {
    wave_sync();
    const uint32_t base = k6_stage(P, g * 64, lin, la0, lane);
    wave_sync();
    const uint64_t sub = g * 64 + (uint64_t)lane;
    if (sub >= P.nsub) continue;
    const uint32_t limit = k6_limit(body_bits, g);
    const uint32_t lo = (uint32_t)lane * kSubBits;
    if (MODE == 2 && CL.L && limit >= 64u * kSubBits) {  // (wave-uniform: every lane's subsequence is whole)
      auto walk = [&](auto l_tag) {
        constexpr int CLEN = decltype(l_tag)::value, NCH = CLEN + 1;  // the chains: entry offsets 0 .. L
        uint64_t F[8], G[8];
        k6_classes<CLEN, true>(lin, la0, base + lo, CL.eof_bit, F, G);
        wave_sync();  // (every lane has its bytes: the tile may take the G masks)
        k6_mask_t* const Fl = k6_f_slot(L, tid);
        k6_mask_t* const Gl = k6_g_slot(L, wave, lane);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          Fl[r * kK6Threads] = F[r];
          Gl[r * 64] = G[r];
        }
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
            // (L = 8: the long code lies in the chain's own class, its G mask is read WITH the F mask -- one LDS round trip per
            //  jump.  L = 4: it lies in one of two classes; reading both candidates' G masks up front gave wrong end-mark bits
            //  in THIS kernel at -O3 -- not at -O1, not with every s_waitcnt forced to zero (so it is not a counted wait), not
            //  in stand-alone GPU harnesses of the same function and loop, scratch/jump_gpu_test.hip / walk_gpu_test.hip, and
            //  not on the host -- which is not root-caused; that build is also the only one that spills (4 VGPRs).  The mask
            //  of the class the jump arrives in is read behind it, and tests/test_cabi_cpu.py keeps this kernel scratch-free.)
            const uint32_t sh = k6_jump<CLEN, CLEN == 8>([&](uint32_t c) { return Fl[c * kK6Threads]; }, [&](uint32_t c) { return Gl[c * 64]; }, pe, found, eof);
            const uint32_t q = pe + (uint32_t)CLEN * sh;
            if (CLEN == 4) eof = found && ((Gl[(q & 7u) * 64] << ((q >> 3) & 63u)) >> 63) != 0;
            p[e] = act ? (found ? q + (uint32_t)CLEN + 1u : q) : p[e];
            nl[e] += (act && found) ? 1u : 0u;
            eofs |= (act && eof) ? (1u << e) : 0u;
            any |= p[e] < 512u;
          }
        }
        uint4 row;
        if (CLEN == 8) {
          uint32_t w3 = 0;
#pragma unroll
          for (int e = 0; e < NCH; ++e) w3 |= (nl[e] < 7u ? nl[e] : 7u) << (3 * e);
          row = make_uint4((p[0] - 512u) | (p[1] - 512u) << 8 | (p[2] - 512u) << 16 | (p[3] - 512u) << 24,
                           (p[4] - 512u) | (p[NCH > 5 ? 5 : 0] - 512u) << 8 | (p[NCH > 6 ? 6 : 0] - 512u) << 16 | (p[NCH > 7 ? 7 : 0] - 512u) << 24,
                           (p[NCH > 8 ? 8 : 0] - 512u) | eofs << 16, w3);
        } else {
          uint32_t n[NCH];
#pragma unroll
          for (int e = 0; e < NCH; ++e) n[e] = nl[e] < 255u ? nl[e] : 255u;
          row = make_uint4((p[0] - 512u) | (p[1] - 512u) << 8 | (p[2] - 512u) << 16 | (p[3] - 512u) << 24,
                           (p[4] - 512u) | n[0] << 8 | n[1] << 16 | n[2] << 24, n[3] | n[4] << 8 | eofs << 16, 0u);
        }
        *reinterpret_cast<uint4*>(tab + sub * stride) = row;
      };
      if (CL.L == 8) walk(std::integral_constant<int, 8>{});
      else walk(std::integral_constant<int, 4>{});
      continue;
    }
    const uint32_t hi = lo + kSubBits < limit ? lo + kSubBits : limit;  // (behind the stream's end nothing is decoded)
    uint8_t* const row = tab + sub * stride;
    constexpr uint32_t NC = 5;  // chains in flight per lane
    const uint32_t nfree = limit >= 64u * kSubBits ? ((uint32_t)kSubBits - S) / (uint32_t)max_len : 0u;  // (wave-uniform: every lane's subsequence is whole)
    for (uint32_t s0 = 0; s0 < S; s0 += NC) {
      K6Cursor c[NC];
      uint32_t p[NC];
#pragma unroll
      for (uint32_t j = 0; j < NC; ++j) {
        p[j] = lo + s0 + j;
        c[j].open(lin, la0, base + p[j]);
        if (s0 + j >= S && nfree == 0u) p[j] = hi;  // no such chain (with free steps it simply runs along: never stored)
      }
      // the first (512 - S) / max_len steps cannot carry any chain out of its subsequence: no test per step (a chain
      // slot that stands for no entry offset steps along behind the subsequence -- inside the tile, never stored)
      for (uint32_t k = 0; k < nfree; ++k) {
#pragma unroll
        for (uint32_t j = 0; j < NC; ++j) p[j] += (k6_step<MODE>(c[j], lin, L, T1, lut_bits, max_len, C2) >> 8) & 0xFFu;
      }
      for (;;) {
        bool any = false;
#pragma unroll
        for (uint32_t j = 0; j < NC; ++j) {
          if (p[j] < hi) {
            p[j] += (k6_step<MODE>(c[j], lin, L, T1, lut_bits, max_len, C2) >> 8) & 0xFFu;
            any = true;
          }
        }
        if (!any) break;
      }
      const uint32_t end = lo + kSubBits;
#pragma unroll
      for (uint32_t j = 0; j < NC; ++j)
        if (s0 + j < S) row[s0 + j] = p[j] >= end ? (uint8_t)(p[j] - end) : (uint8_t)0;
    }
  }
  };
  if (k6_mode == 2) run(std::integral_constant<int, 2>{});
  else if (k6_mode == 1) run(std::integral_constant<int, 1>{});
  else run(std::integral_constant<int, 0>{});
}

// one level up: out[t] = f[64 t + 63] o ... o f[64 t]  (entry s: where a decode that enters tile t at offset s leaves it)
__global__ __launch_bounds__(64) void k_fn_reduce(const uint8_t* __restrict__ f, uint64_t n, uint32_t stride, uint8_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint8_t fl[64 * 64];
  const uint64_t t = blockIdx.x;
  const int lane = threadIdx.x;
  const uint64_t first = t * 64;
  const int cnt = (int)((n - first < 64) ? (n - first) : 64);
  for (int i = lane; i < cnt * (int)stride / 16; i += 64)
    reinterpret_cast<uint4*>(fl)[i] = reinterpret_cast<const uint4*>(f + first * stride)[i];
  __syncthreads();
  if (lane < (int)stride) {
    uint32_t cur = (uint32_t)lane;
    for (int j = 0; j < cnt; ++j) cur = fl[j * stride + (cur & (stride - 1))];
    out[t * stride + lane] = (uint8_t)cur;
  }
}

// one level down: start[64 t + j] = offset at which the true decode enters element j of tile t, given where it enters the
// tile.  The lowest level writes the subsequences' start offsets themselves (16-bit, P.start; [0] keeps the caller's value).
// With classes (*cls_flag = L, see k_sync_table) the lowest level also SETTLES the subsequences below n_settle: count and
// "computed from this entry offset" as k_sync_pass would leave them, unless the row says that the chain from this entry
// met an end mark or more 9-bit codes than the row counts.
template <typename OutT>
__global__ __launch_bounds__(64) void k_fn_apply(const uint8_t* __restrict__ f, uint64_t n, uint32_t stride,
                                                 const uint8_t* __restrict__ tile_start, uint32_t entry, OutT* __restrict__ start,
                                                 const uint32_t* __restrict__ cls_flag, uint64_t n_settle, uint32_t* __restrict__ cnt_out,
                                                 uint16_t* __restrict__ used_out, uint8_t* __restrict__ eof_out) {
  __shared__ __attribute__((aligned(16))) uint8_t fl[64 * 64];
  __shared__ uint8_t st[64];
  const uint64_t t = blockIdx.x;
  const int lane = threadIdx.x;
  const uint64_t first = t * 64;
  const int cnt = (int)((n - first < 64) ? (n - first) : 64);
  for (int i = lane; i < cnt * (int)stride / 16; i += 64)
    reinterpret_cast<uint4*>(fl)[i] = reinterpret_cast<const uint4*>(f + first * stride)[i];
  __syncthreads();
  if (lane == 0) {
    uint32_t cur = tile_start ? tile_start[t] : entry;  // top level: where the caller says the first code begins
    for (int j = 0; j < cnt; ++j) {
      st[j] = (uint8_t)cur;
      cur = fl[j * stride + (cur & (stride - 1))];
    }
  }
  __syncthreads();
  if (lane < cnt && (sizeof(OutT) == 1 || first + lane != 0)) start[first + lane] = (OutT)st[lane];
  const uint32_t clen = (sizeof(OutT) == 2 && cls_flag) ? *cls_flag : 0u;
  if (clen && lane < cnt && first + lane < n_settle) {
    const uint32_t e = st[lane];
    const uint8_t* const row = fl + lane * stride;
    const uint32_t eofs = (*reinterpret_cast<const uint16_t*>(row + 10) >> e) & 1u;
    uint32_t nl, full;  // (L + 1)-bit codes on the chain from e; the value that stands for "or more"
    if (clen == 8u) {
      nl = (*reinterpret_cast<const uint32_t*>(row + 12) >> (3u * e)) & 7u;
      full = 7u;
    } else {
      nl = row[5u + (e < 5u ? e : 0u)];
      full = 255u;
    }
    if (e <= clen && nl != full && !eofs) {  // bits from the entry offset to the landing bit = L per code + 1 per long code
      cnt_out[first + lane] = ((uint32_t)kSubBits + row[e] - e - nl) / clen;
      used_out[first + lane] = (uint16_t)e;
      eof_out[first + lane] = 0;  // (a pass that ran on an earlier guess may have seen a fake end mark here)
    }
  }
}

size_t sync_scan_workspace(uint64_t nsub) {
  uint64_t n = nsub, total = 0;
  for (;;) {
    total += n;
    if (n <= 1) break;
    n = (n + 63) / 64;
  }
  return (size_t)(total * (64 + 1) + 256 * 33);  // functions (<= 64 bytes) + entry offsets of every level, each level 256-aligned, + the class flag
}

static uint32_t k6_blocks(uint64_t nsub) {
  const uint64_t groups = (nsub + 63) / 64;
  uint64_t blocks = (groups + kK6Waves - 1) / kK6Waves;
  if (blocks > 256) blocks = 256;  // one workgroup per CU (LDS)
  return blocks ? (uint32_t)blocks : 1u;
}

void launch_sync_scan(const SyncParams& p, uint8_t* ws, uint32_t stride, uint32_t entry, hipStream_t s) {
  if (p.nsub == 0) return;
  // carve: functions of level 0.., then entry offsets of level 1..
  uint64_t cnt[16];
  uint8_t* fn[16];
  uint8_t* st[16];
  int levels = 0;
  uint8_t* q = ws;
  for (uint64_t n = p.nsub;; n = (n + 63) / 64) {
    cnt[levels] = n;
    fn[levels] = q;
    q += (n * stride + 255) & ~(uint64_t)255;
    ++levels;
    if (n <= 1 || levels == 16) break;
  }
  for (int l = 0; l < levels; ++l) {
    st[l] = q;
    q += (cnt[l] + 255) & ~(uint64_t)255;
  }
  uint32_t* const cls_flag = reinterpret_cast<uint32_t*>(q);
  hipLaunchKernelGGL(k_sync_table, dim3(k6_blocks(p.nsub)), dim3(kK6Threads), 0, s, p, stride, fn[0], cls_flag);
  for (int l = 0; l + 1 < levels; ++l)
    hipLaunchKernelGGL(k_fn_reduce, dim3((uint32_t)cnt[l + 1]), dim3(64), 0, s, fn[l], cnt[l], stride, fn[l + 1]);
  // the top level has one element: the whole body, entered at bit `entry` (< stride) of its first subsequence
  if (levels == 1) {
    return;  // a single subsequence: its start is the caller's
  }
  const uint32_t* const no_flag = nullptr;
  hipLaunchKernelGGL(k_fn_apply<uint8_t>, dim3(1), dim3(64), 0, s, fn[levels - 1], cnt[levels - 1], stride, (const uint8_t*)nullptr,
                     entry, st[levels - 1], no_flag, 0ull, (uint32_t*)nullptr, (uint16_t*)nullptr, (uint8_t*)nullptr);
  for (int l = levels - 2; l >= 1; --l)
    hipLaunchKernelGGL(k_fn_apply<uint8_t>, dim3((uint32_t)cnt[l + 1]), dim3(64), 0, s, fn[l], cnt[l], stride, st[l + 1], 0u, st[l], no_flag,
                       0ull, (uint32_t*)nullptr, (uint16_t*)nullptr, (uint8_t*)nullptr);
  // settled by the lowest level (class codes only): whole groups of 64 subsequences, and never the last subsequence (its
  // landing bit, the end of a piece and the end mark are k_sync_pass's)
  uint64_t n_settle = (p.end_bit - p.body_bit0) / (64ull * kSubBits) * 64ull;
  if (n_settle > p.nsub - 1) n_settle = p.nsub - 1;
  hipLaunchKernelGGL(k_fn_apply<uint16_t>, dim3((uint32_t)cnt[1]), dim3(64), 0, s, fn[0], cnt[0], stride, st[1], 0u, p.start,
                     (const uint32_t*)cls_flag, (unsigned long long)n_settle, p.cnt, p.used, p.eof);
}

// first subsequence that holds the end mark (valid once the passes have converged)
__global__ __launch_bounds__(256) void k_sync_eof(SyncParams P) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < P.nsub && P.eof[i]) atomicMin(reinterpret_cast<unsigned long long*>(P.eof_sub), (unsigned long long)i);
}

// symbols per tile of 256 subsequences, nothing counted behind the end mark
__global__ __launch_bounds__(256) void k_sync_tile_sums(SyncParams P) {
  __shared__ unsigned long long ws[4];
  const uint64_t eof_sub = *P.eof_sub;
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned long long v = (i < P.nsub && i <= eof_sub) ? P.cnt[i] : 0ull;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) P.tile_sum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// absolute bit position of every 64th symbol (the side-car's granularity); a workgroup trip = 1024 subsequences = four of
// the 256-subsequence tiles whose symbol counts k_sync_tile_sums / k_scan prefix-summed
__global__ __launch_bounds__(kK6Threads, 4) void k_sync_index(SyncParams P, uint64_t* __restrict__ seg_abs, uint64_t n_segs, uint64_t n_symbols) {
  GHF_K6_PROLOGUE();
  (void)ngroups;
  const K6Cls CL = k6_cls(k6_mode, P.dt->min_len, max_len, C2, P.body_bit0);
  __shared__ unsigned long long wsum[kK6Waves];
  const uint64_t eof_sub = *P.eof_sub;
  const uint64_t ntrips = (P.nsub + kK6Threads - 1) / kK6Threads;
  auto run = [&](auto mode_tag) {
  constexpr int MODE = decltype(mode_tag)::value;
  for (uint64_t trip = blockIdx.x; trip < ntrips; trip += gridDim.x) {
    const uint64_t g = trip * kK6Waves + wave;  // this wave's group of 64 subsequences
    const uint64_t sub = g * 64 + lane;
    const bool valid = sub < P.nsub && sub <= eof_sub;
    const uint32_t c = valid ? P.cnt[sub] : 0u;
    // exclusive prefix of the symbol counts inside the 256-subsequence tile (4 waves)
    const uint32_t incl = wave_incl_scan_u32(c);
    __syncthreads();  // (the previous trip's readers of wsum are done)
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned long long first = incl - c;
    if (g * 64 < P.nsub) {
      first += P.tile_sum[g >> 2];  // tile_sum[] holds the exclusive scan by now
      for (int k = wave & ~3; k < wave; ++k) first += wsum[k];
    }
    if (g * 64 >= P.nsub) continue;  // (uniform per wave; the barriers above were passed by everyone)
    wave_sync();
    const uint32_t base = k6_stage(P, g * 64, lin, la0, lane);
    wave_sync();
    const uint32_t limit = k6_limit(body_bits, g);
    const uint64_t abs0 = P.body_bit0 + g * 64 * kSubBits;  // stream bit of the wave's first subsequence
    if (MODE == 2 && CL.L && limit >= 64u * kSubBits) {  // class codes (wave-uniform): the true chain as a walk over the class masks
      auto walk = [&](auto l_tag) {
        constexpr int CLEN = decltype(l_tag)::value;
        uint64_t F[8], G[8];
        k6_classes<CLEN, false>(lin, la0, base + (uint32_t)lane * kSubBits, 0u, F, G);
        k6_mask_t* const Fl = k6_f_slot(L, tid);
#pragma unroll
        for (int r = 0; r < 8; ++r) Fl[r * kK6Threads] = F[r];
        if (!valid || c == 0u) return;
        // symbol numbers m0 (and, L = 4, m0 + 64) of the chain are the first of a segment, symbol number c the first one
        // that is not counted here
        const uint32_t m0 = (uint32_t)((64u - (uint32_t)(first & 63u)) & 63u);
        const uint32_t lo = (uint32_t)lane * kSubBits;
        uint32_t p = P.start[sub], k0 = 0, p_end = 0;
        while (k0 < c) {  // one run of L-bit codes, closed by an (L + 1)-bit code or by the end of the subsequence
          bool found, eof;
          const uint32_t sh = k6_jump<CLEN, false>([&](uint32_t c) { return Fl[c * kK6Threads]; }, [](uint32_t) { return 0ull; }, p, found, eof);
          const uint32_t run = found ? sh + 1u : sh;  // codes in it
#pragma unroll
          for (uint32_t m = m0; m < 64u * (8u / (uint32_t)CLEN); m += 64u) {
            const uint64_t seg = (first + m) >> 6;
            if (m >= k0 && m < k0 + run && m < c && seg < n_segs) seg_abs[seg] = abs0 + lo + p + (uint32_t)CLEN * (m - k0);
          }
          if (c <= k0 + run) p_end = p + (uint32_t)CLEN * (c - k0) + ((found && c == k0 + run) ? 1u : 0u);
          p += (uint32_t)CLEN * sh + (uint32_t)CLEN + 1u;
          k0 += run;
          if (!found) break;  // (the chain has left the subsequence: c <= k0 by the count's definition)
        }
        if (first + c == n_symbols) seg_abs[n_segs] = abs0 + lo + p_end;  // where the last data symbol ends
      };
      if (CL.L == 8) walk(std::integral_constant<int, 8>{});
      else walk(std::integral_constant<int, 4>{});
      continue;
    }
    if (!valid) continue;
    uint32_t pos = (uint32_t)lane * kSubBits + P.start[sub];
    K6Cursor cur;
    cur.open(lin, la0, base + pos);
    uint32_t mark = (uint32_t)((64u - (uint32_t)(first & 63u)) & 63u);  // my symbols in front of the next segment start
    uint64_t seg = (first + mark) >> 6;
    const bool pairs = MODE != 2 && max_len <= 16 && limit >= 64u * kSubBits;  // (wave-uniform; see k_sync_pass)
    for (uint32_t k = 0; k < c && pos < limit;) {
      if (k == mark) {
        if (seg < n_segs) seg_abs[seg] = abs0 + pos;
        ++seg;
        mark += 64u;
      }
      const uint32_t stop = c < mark ? c : mark;  // the next symbol number at which something is written
      if (pairs) {  // (the counts are exact: no end mark among these symbols, none of them behind the stream's end)
        for (; k + 2u <= stop; k += 2u) {
          uint32_t e0, e1;
          k6_peek2<MODE == 2 ? 1 : MODE>(cur, lin, L, T1, lut_bits, max_len, e0, e1);
          const uint32_t l = ((e0 >> 8) & 0xFFu) + ((e1 >> 8) & 0xFFu);
          pos += l;
          cur.o += l;
        }
      }
      if (k < stop) {
        pos += (k6_step<MODE>(cur, lin, L, T1, lut_bits, max_len, C2) >> 8) & 0xFFu;
        ++k;
      }
    }
    if (c && first + c == n_symbols) seg_abs[n_segs] = abs0 + pos;  // where the last data symbol ends
  }
  };
  if (k6_mode == 2) run(std::integral_constant<int, 2>{});
  else if (k6_mode == 1) run(std::integral_constant<int, 1>{});
  else run(std::integral_constant<int, 0>{});
}

// seg_abs[s] = stream bit of symbol 64 s (s < n_segs), seg_abs[n_segs] = end of the last symbol  ->  the side-car K5 emits:
// absolute start bit per block of 64 segments, end bit of every segment relative to its block
__global__ __launch_bounds__(256) void k_sync_finalize(const uint64_t* __restrict__ seg_abs, uint64_t n_segs,
                                                       uint64_t* __restrict__ chunk_bit, uint32_t* __restrict__ seg_bit) {
  const uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= n_segs) return;
  const uint64_t b = s / (kBlockSymbols / kSegSymbols);
  const uint64_t b0 = seg_abs[b * (kBlockSymbols / kSegSymbols)];
  if (s == b * (kBlockSymbols / kSegSymbols)) chunk_bit[b] = b0;
  seg_bit[s] = (uint32_t)(seg_abs[s + 1] - b0);
}

void launch_sync_pass(const SyncParams& p, hipStream_t s) {
  hipLaunchKernelGGL(k_sync_pass, dim3(k6_blocks(p.nsub)), dim3(kK6Threads), 0, s, p);
}
void launch_sync_counts(const SyncParams& p, uint64_t* d_total, hipStream_t s) {
  const uint32_t tiles = (uint32_t)((p.nsub + 255) / 256);
  hipLaunchKernelGGL(k_sync_eof, dim3(tiles), dim3(256), 0, s, p);
  hipLaunchKernelGGL(k_sync_tile_sums, dim3(tiles), dim3(256), 0, s, p);
  launch_scan(p.tile_sum, tiles, d_total, s);
}
void launch_sync_index(const SyncParams& p, uint64_t* d_seg_abs, uint64_t n_symbols, uint64_t* d_chunk_bit, uint32_t* d_seg_bit,
                       hipStream_t s) {
  const uint64_t n_segs = (n_symbols + kSegSymbols - 1) / kSegSymbols;
  const uint64_t trips = (p.nsub + kK6Threads - 1) / kK6Threads;
  hipLaunchKernelGGL(k_sync_index, dim3((uint32_t)(trips < 256 ? (trips ? trips : 1) : 256)), dim3(kK6Threads), 0, s, p, d_seg_abs, n_segs, n_symbols);
  if (n_segs) hipLaunchKernelGGL(k_sync_finalize, dim3((uint32_t)((n_segs + 255) / 256)), dim3(256), 0, s, d_seg_abs, n_segs,
                                 d_chunk_bit, d_seg_bit);
}

}  // namespace ghf
