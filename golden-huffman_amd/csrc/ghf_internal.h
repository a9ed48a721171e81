// golden-huffman_amd/csrc/ghf_internal.h -- shared between the HIP kernels and the C-ABI host layer.
#ifndef GHF_INTERNAL_H_
#define GHF_INTERNAL_H_
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ghf.h"

namespace ghf {

// ---- geometry ---------------------------------------------------------------------------------
constexpr int kWave = 64;               // CDNA wavefront
constexpr int kSymPerLane = 16;         // one 16-byte vector load = one lane's contiguous symbols
constexpr int kSymPerIter = kWave * kSymPerLane;  // 1 KiB of input per wave iteration
constexpr int kSegSymbols = 64;         // side-car granularity (4 lanes of K5, one lane of K7)
constexpr int kBlockSymbols = 4096;     // side-car block = 64 segments = one K7 group = 4 K5 tiles; absolute bit per block
// A chunk = the input one K5 wave packs (and the unit K1 prices for K4).  Chunks are sized so that ALL of them are resident
// at once: K5 keeps 3 workgroups x 8 waves on each of the 256 CUs, and a grid of 1.33 rounds (8192 power-of-two chunks, as in
// round 1) spends its last third with a third of the waves -- too few loads in flight to keep HBM busy.  Multiples of
// 4 KiB (K1's vector rows: 256 threads x 16 bytes; K5's trips of four 1 KiB tiles), at least 16 KiB, at most 1 MiB.  (Round 3 tried 2 workgroups x 16 waves, 8 waves per SIMD:
// 3 % faster alone, 12 % slower between the other kernels of the pipelined bench -- a 1024-thread workgroup needs half a
// CU to drain before it can start.)
constexpr uint32_t kChunkQuantum = 4096;
constexpr uint32_t kMinChunk = 16384;
constexpr uint32_t kMaxChunk = 1u << 20;
constexpr uint32_t kEmitSlots = 256 * 3 * 8;

// K1 histogram
constexpr int kHistThreads = 256;
constexpr int kHistRep = 32;  // bins[256][32]: replica = lane % 32 -> every lane of a 32-lane LDS group owns a bank

// K5 emit
constexpr int kEmitThreads = 512;                 // 8 waves share one 32 KiB replicated code table
constexpr int kEmitWaves = kEmitThreads / kWave;
constexpr int kStageWords = 544;                  // per-wave staging: 128 carry bits + 16384 bits + slack
constexpr int kStageCapBits = (kStageWords - 8) * 32;

// K7 decode / K6 side-car reconstruction
constexpr int kDecPairBitsMax = 10;
constexpr int kDecLutBitsMax = 12;
// K7 / K6 keep the direct table in LDS as 32-bit entries, replicated (ghf_decode.hip): 64 KiB of slots + a small region
constexpr int kDec7LutLog2 = 14;
constexpr int kDec7LutSlots = 1 << kDec7LutLog2;
constexpr int kDec7SmallSlots = 1024;  // pair mode (max_len <= 5): the one-symbol table for ragged tails and the end mark, 32 copies

inline uint32_t chunk_symbols_for(uint64_t n) {
  const uint64_t per_slot = (n + kEmitSlots - 1) / kEmitSlots;
  uint64_t c = (per_slot + kChunkQuantum - 1) / kChunkQuantum * kChunkQuantum;
  if (c < kMinChunk) c = kMinChunk;
  if (c > kMaxChunk) c = kMaxChunk;
  return (uint32_t)c;
}
inline uint64_t chunk_count_for(uint64_t n) {
  const uint64_t c = chunk_symbols_for(n);
  return (n + c - 1) / c;
}

// decode-side tables derived from ghf_code (built on the device by k_build_decode_tables)
struct DecTables {
  uint32_t fc_left[36];    // first_code[len] << (32 - len), len = 1..max_len; 0xFFFFFFFF for len < min_len
  uint32_t start_pos[36];
  uint16_t symbol[GHF_NSYM + 3];
  int32_t min_len, max_len, lut_bits;
  int32_t pair_bits;       // 2 * max_len when that is <= kDecPairBitsMax (every pair of codes fits lut2's index), else 0
  // kind 1 (.crs): the codes are whatever the tree says; codes beyond the direct table are decoded by
  // walking tl/tr (children of parent i; ids < 256 = leaf key, else 256 + parent index) from `root`
  int32_t kind;
  uint32_t root;
  uint16_t tl[256], tr[256];
  // work counters of k_decode, zeroed by k_build_decode_tables.  One word saturates at ~88 tickets/us (measured:
  // a single counter made the 256 MiB decode 4.8x slower), so the workgroups are split into 16 classes
  // (blockIdx % 16), each with its own counter on its own 128-byte line; class c owns the groups == c (mod 16).
  uint32_t ticket[16 * 32];
  // workgroups of the running k_decode that have finished; the last one zeroes the counters again (and this word), so
  // that tables stay usable for any number of launches
  uint32_t done;
  // The direct table(s) exactly as K7 / K6 keep them in LDS (ghf_decode.hip: 32-bit entries, replicated): written once by the
  // table kernels, pulled in by every workgroup with 16-byte loads (round 3 replicated a compact table in every workgroup's
  // prologue: ~3 us per launch of K7 and of every K6 kernel).  One-symbol table: index = the next lut_bits stream bits,
  //   entry = symbol | length << 8 | bit 16: end mark | bit 17: no code of <= lut_bits bits starts with these bits
  // in min(32, 2^(14 - lut_bits)) copies, slot = index * copies + copy.  Small alphabets (pair_bits != 0): the first 2^14 slots
  // hold the pair table -- index = the next pair_bits bits, entry = sym0 | sym1 << 8 | (len0 + len1) << 16 | bit 30: not two data
  // symbols -- and the one-symbol table follows in 32 copies.
  alignas(16) uint32_t image[kDec7LutSlots + kDec7SmallSlots];
};

struct EmitParams {
  const uint8_t* in;
  uint64_t n;
  const ghf_code* code;
  const uint64_t* chunk_off;  // [nchunks + 1] bits relative to this buffer's first code (exclusive scan)
  const uint64_t* d_start_bit;
  uint8_t* out;
  uint64_t cap;
  uint32_t chunk;       // symbols per chunk (a multiple of 4 KiB, >= 16 KiB)
  uint32_t nchunks;
  uint64_t* chunk_bit;  // side-car (may be null): [n / 4096] absolute start bit of every block
  uint32_t* seg_bit;    // side-car (may be null): [n / 64] end bit of every segment, relative to its block
  int flags;
  int* status;
  uint64_t* d_end;  // may be null
};

struct DecParams {
  const uint8_t* stream;
  uint64_t stream_bytes;
  DecTables* dt;
  const uint64_t* chunk_bit;
  const uint32_t* seg_bit;
  uint64_t n_symbols;
  uint64_t n_segs;
  uint32_t no_end_mark;  // the last symbol is not followed by the end mark (a shard that is not the stream's last)
  uint8_t* out;
  uint64_t* out_bytes;  // optional device u64 <- n_symbols
  int* status;
};

// K6: side-car reconstruction for foreign streams
struct SyncParams {
  const uint8_t* stream;
  uint64_t stream_bytes;
  uint64_t body_bit0;  // first body bit = 8 * header bytes
  uint64_t end_bit;    // one past the last bit that may belong to a code (8 * stream_bytes for .crs2)
  uint32_t no_eof;     // 0: .crs2, ends with the end mark.  1: .crs, no end mark, the last code must end exactly at end_bit.
                       // 2: a piece of a .crs2 (multi-GPU decode): an end mark ends it if there is one, otherwise the last
                       //    code may run past end_bit and start[nsub] receives by how much
  const DecTables* dt;
  uint64_t nsub;       // 512-bit subsequences covering the body
  uint16_t* start;     // [nsub + 1] current guess: bit offset of the first code boundary inside each subsequence
  uint16_t* used;      // [nsub]     the guess the stored result was computed from (0xFFFF = none yet)
  uint32_t* cnt;       // [nsub]     codes starting in the subsequence (up to an end mark)
  uint8_t* eof;        // [nsub]     the end mark was decoded in this subsequence
  uint32_t first;      // k_sync_pass only.  bit 0: every subsequence has work (`used`, `eof` hold nothing yet: the launch that
                       // writes them all needs no memset in front of it); bit 1: every guess is 0 except subsequence 0's
                       // (`start` holds nothing yet either; this launch writes start[1 .. nsub])
  uint32_t first_start;  // subsequence 0's guess (bit 1 of `first`)
  uint32_t* changed;   // [0] some guess moved during the pass; [1] how many did, roughly (every 256th group counts)
  unsigned long long* moved_first_inv;  // ~(smallest subsequence whose landing moved during the pass); 0: none (sits behind `changed`)
  uint64_t* eof_sub;   // first subsequence holding the end mark
  uint64_t* tile_sum;  // [nsub / 256 + 2] symbols per tile, then (in place) their exclusive scan
};

// kernel launchers (ghf_kernels.hip); all asynchronous on `s`
void launch_sync_pass(const SyncParams& p, hipStream_t s);
void launch_sync_counts(const SyncParams& p, uint64_t* d_total, hipStream_t s);
// deterministic seeding of p.start[] (function-composition scan); ws = sync_scan_workspace(p.nsub) bytes, 256-byte aligned
size_t sync_scan_workspace(uint64_t nsub);
void launch_sync_scan(const SyncParams& p, uint8_t* ws, uint32_t stride /* 16: max_len <= 16 is known; 64: it may exceed 32 (.crs); else 32 */,
                      uint32_t entry /* bit at which the first code begins, < stride */, hipStream_t s);
void launch_sync_index(const SyncParams& p, uint64_t* d_seg_abs, uint64_t n_symbols, uint64_t* d_chunk_bit, uint32_t* d_seg_bit,
                       hipStream_t s);
// K1 scratch, all zero between launches: 32 replicas of the 256 totals, the arrival counter (word 8192), 16 ticket
// counters (word 8208 + 16 k, one 128-byte line each)
constexpr size_t kHistAccWords = 32 * 256 + 16 + 16 * 16;
void launch_histogram(const uint8_t* d_in, uint64_t n, uint32_t chunk, uint32_t nchunks, uint32_t* d_chunk_hist,
                      uint64_t* d_hist, uint64_t* d_acc, bool add, hipStream_t s);
void launch_build_code(const uint64_t* d_hist, ghf_code* d_code, int* d_status, uint32_t flags, hipStream_t s);
void launch_write_header(const ghf_code* d_code, uint8_t* d_out, uint64_t cap, int* d_status, hipStream_t s);
void launch_plan(const uint8_t* d_in, uint64_t n, uint32_t chunk, uint32_t nchunks, const uint32_t* d_chunk_hist,
                 const ghf_code* d_code, uint64_t* d_chunk_off, uint64_t* d_total_bits, hipStream_t s);
// in-place exclusive scan of d_v[0..count), d_v[count] = total, *d_total = total (one workgroup)
void launch_scan(uint64_t* d_v, uint32_t count, uint64_t* d_total, hipStream_t s);
void launch_emit(const EmitParams& p, hipStream_t s);
void launch_build_decode_tables(const ghf_code* d_code, DecTables* d_dt, int* d_status, hipStream_t s);
constexpr uint64_t kDecMaxGroups = 0xFFFF0000ull;  // groups of 4096 symbols one k_decode launch takes (2^44 symbols)
void launch_decode(const DecParams& p, hipStream_t s);
void launch_crs_build_code(const uint64_t* d_hist, ghf_tree* d_tree, ghf_code* d_code, uint64_t* d_start_bit, int* d_status,
                           hipStream_t s);
void launch_crs_finish(const ghf_tree* d_tree, const uint64_t* d_total_bits, uint8_t* d_out, uint64_t* d_out_bytes, int* d_status,
                       hipStream_t s);
void launch_crs_decode_tables(const ghf_tree* d_tree, DecTables* d_dt, int* d_status, hipStream_t s);
void launch_stream_copy(const uint8_t* d_src, uint8_t* d_dst, uint64_t n, bool nt, hipStream_t s);
void launch_store_u64(uint64_t* d_dst, const uint64_t* d_src_opt, uint64_t add, hipStream_t s);
void launch_load_u16(uint64_t* d_dst, const uint16_t* d_src, hipStream_t s);
void launch_shard_start(const ghf_code* d_code, const uint64_t* d_totals, int rank, uint64_t* d_start_bit, hipStream_t s);

}  // namespace ghf
#endif
