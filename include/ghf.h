/* include/ghf.h -- C ABI of libghf.so: the MI355X (gfx950) canonical-Huffman hot path.
 *
 * Drop-in boundary for the byte-keyed canonical-Huffman path of chenghuige/golden-huffman (`glzip`).
 * Each entry point names the reference interface it replaces (file:line relative to the reference
 * tree).  Plain pointers and sizes only; no C++/torch/HIP types cross this boundary (a HIP stream is
 * passed as void*).  Every pointer whose name starts with d_ is DEVICE memory on the context's GPU
 * and must be 16-byte aligned unless stated otherwise; everything else is host memory.
 *
 * All stage calls are asynchronous on the context's stream and never synchronise with the host.
 * Device-side failures (code longer than 32 bits, empty input, output capacity exceeded) are latched
 * in a device status word; ghf_sync() / ghf_status() return them.  No call throws.
 *
 * The product path has no CPU fallback: without a HIP device every stage call returns GHF_E_HIP.
 */
#ifndef GHF_H_
#define GHF_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GHF_NSYM 257 /* include/type_traits.h:50 CharSymbolNum = 256 byte values + end-of-stream mark */

enum ghf_status_code {
  GHF_OK = 0,
  GHF_E_INVAL = 1,   /* bad argument (null, misaligned, ...) */
  GHF_E_HIP = 2,     /* HIP runtime error / no device; see ghf_last_error() */
  GHF_E_EMPTY = 3,   /* n == 0: the reference is undefined there; we refuse */
  GHF_E_CODELEN = 4, /* a code longer than 32 bits (reference limit, include/canonical_huff_encoder.h:43-44) */
  GHF_E_CAP = 5,     /* output capacity too small */
  GHF_E_FORMAT = 6,  /* not a .crs2 header */
  GHF_E_CORRUPT = 7, /* stream does not decode to the expected symbol count / end mark */
  GHF_E_NOMEM = 8,
  GHF_E_SINGLE = 9 /* .crs only: one distinct byte value -- the lone leaf gets the empty code and the reference's
                      decoder dereferences a NULL child (include/huff_tree.cc:255-271); undefined there, refused here */
};

/* The encoder's tables -- mirrors the private members of CanonicalHuffEncoder,
 * include/canonical_huff_encoder.h:107-120 (length_, codeword_, symbol_, first_code_, start_pos_,
 * min_len_, max_len_).  first_code/start_pos are indexed from 1 like the reference's. */
typedef struct ghf_code {
  uint32_t length[GHF_NSYM];
  uint32_t codeword[GHF_NSYM];
  uint32_t symbol[GHF_NSYM]; /* unused tail = 0xFFFFFFFF (canonical_huff_encoder.cc:88) */
  uint32_t first_code[64];   /* 1024 for len < min_len (canonical_huff_encoder.cc:119-121) */
  uint32_t start_pos[64];
  int32_t min_len;
  int32_t max_len;
} ghf_code;

/* Side-car index for block-parallel decode (no reference counterpart: the .crs2 wire format has no
 * sync points).  It is NOT part of the .crs2 bytes.  The struct lives on the host;
 * the two arrays are device memory (ghf_index_alloc). */
typedef struct ghf_index {
  uint64_t n_symbols;     /* input bytes covered */
  uint32_t chunk_symbols; /* symbols per index block (4096 = 64 segments) */
  uint32_t seg_symbols;   /* symbols per segment (64) */
  uint64_t n_chunks;      /* index blocks */
  uint64_t n_segs;
  uint32_t flags;    /* GHF_INDEX_NO_END_MARK: the buffer is a shard that was emitted without GHF_EMIT_LAST */
  uint32_t reserved;
  uint64_t* d_chunk_bit; /* [n_chunks] bit offset of each block's first code, counted from byte 0 of the d_out the
                            emit call wrote into (= absolute stream bit, header included, unless GHF_EMIT_REBASE) */
  uint32_t* d_seg_bit;   /* [n_segs]   bit offset of each segment's END relative to its block's first code (a segment
                            starts where its predecessor in the block ends; the first one at the block's first code) */
} ghf_index;

#define GHF_INDEX_NO_END_MARK 1u

typedef struct ghf_ctx ghf_ctx;

/* ---- context (the reference has none: single-threaded objects; SURVEY 8b "Threading") ---------- */
int ghf_ctx_create(int device, ghf_ctx** out);
int ghf_ctx_destroy(ghf_ctx* ctx);
/* A new context owns a private non-blocking stream.  ghf_ctx_set_stream makes it queue on the caller's
 * hipStream_t instead (NULL = HIP's default stream), e.g. torch's current stream. */
int ghf_ctx_set_stream(ghf_ctx* ctx, void* hip_stream);
int ghf_sync(ghf_ctx* ctx);                             /* wait for the stream; returns latched device status */
int ghf_status(ghf_ctx* ctx);                           /* = ghf_sync */
int ghf_clear_status(ghf_ctx* ctx);
const char* ghf_last_error(ghf_ctx* ctx);
const char* ghf_status_string(int status);
int ghf_version(void);

/* ---- memory helpers, so that C/C++ hosts need no HIP headers.  They replace the 64 KiB stdio
 *      buffers of utils/include/buffer.h:61-317 with pinned-host + hipMemcpyAsync staging. ------- */
int ghf_device_alloc(ghf_ctx* ctx, size_t bytes, void** d_ptr);
int ghf_device_free(ghf_ctx* ctx, void* d_ptr);
int ghf_host_alloc(ghf_ctx* ctx, size_t bytes, void** h_ptr); /* pinned */
int ghf_host_free(ghf_ctx* ctx, void* h_ptr);
int ghf_copy_h2d(ghf_ctx* ctx, void* d_dst, const void* h_src, size_t bytes); /* async on the stream */
int ghf_copy_d2h(ghf_ctx* ctx, void* h_dst, const void* d_src, size_t bytes); /* async on the stream */
int ghf_memset_d(ghf_ctx* ctx, void* d_dst, int value, size_t bytes);
/* Device-to-device copy by a kernel with this path's own access shape (16 bytes per lane, four loads in
 * flight, one resident round of workgroups; non_temporal != 0: `global_load/store ... nt`, the hint K1 and K7
 * stream with).  Both pointers 16-byte aligned.  No reference counterpart: it is the bandwidth probe bench.py
 * prices the codec kernels against (what a kernel that only moves the bytes reaches on this box), async on
 * the stream. */
int ghf_copy_d2d(ghf_ctx* ctx, void* d_dst, const void* d_src, size_t bytes, int non_temporal);

/* ---- events: ordering between the streams of several contexts without stopping the host ----------
 * (the file pipeline of golden-huffman_amd/host/glzip_hip.h, which replaces the reference's 64 KiB
 * Buffer, utils/include/buffer.h:61-317, keeps one context per direction: copy-in, kernels, copy-out).
 * record: the event completes when everything enqueued on ctx's stream so far has.  wait: ctx's stream
 * does not run anything enqueued after this call before the event has completed (no host wait).
 * sync: the host waits.  An event that was never recorded counts as complete. */
typedef struct ghf_event ghf_event;
int ghf_event_create(ghf_ctx* ctx, ghf_event** out);
int ghf_event_destroy(ghf_event* ev);
int ghf_event_record(ghf_ctx* ctx, ghf_event* ev);
int ghf_event_wait(ghf_ctx* ctx, ghf_event* ev);
int ghf_event_sync(ghf_event* ev);

/* ---- K1: Encoder::do_init + do_caculate_frequency, include/encoder.h:123-129,136-150 ------------
 * d_hist[0..255] = byte counts of d_in[0..n), d_hist[256] = 1.  d_in needs no alignment (16-byte
 * aligned input takes the fast path).  Also leaves per-chunk histograms in the context so that a
 * following ghf_encode_plan on the same (d_in, n) does not re-read the input -- the caller must not
 * change d_in[0..n) between the two calls. */
int ghf_histogram(ghf_ctx* ctx, const uint8_t* d_in, size_t n, uint64_t* d_hist);
/* The same, for an input that arrives in pieces (the reference counts while it refills its buffer,
 * include/encoder.h:136-150): d_hist[0..255] += byte counts of d_in[0..n), d_hist[256] = 1.  The caller
 * zeroes d_hist (ghf_memset_d) before the first piece.  Keeps nothing for ghf_encode_plan (a piece's
 * buffer is usually refilled in between). */
int ghf_histogram_add(ghf_ctx* ctx, const uint8_t* d_in, size_t n, uint64_t* d_hist);

/* ---- K2+K3: CanonicalHuffEncoder::gen_encode = get_encoding_length + do_gen_encode,
 *      include/canonical_huff_encoder.cc:35-42,289-345,69-141 -- on ONE wavefront, emulating
 *      libstdc++'s priority_queue order exactly.  d_hist is not modified. */
int ghf_build_code(ghf_ctx* ctx, const uint64_t* d_hist, ghf_code* d_code);

/* ---- a5: CanonicalHuffEncoder::write_encode_info, include/canonical_huff_encoder.cc:210-242 ------
 * Writes the 1040 + 8*max_len header bytes (big-endian u32, utils/include/buffer.h:255-268) at d_out. */
int ghf_write_header(ghf_ctx* ctx, const ghf_code* d_code, uint8_t* d_out, size_t cap);
size_t ghf_header_bytes(int max_len); /* 1040 + 8*max_len */

/* ---- K4 + K5: CanonicalHuffEncoder::encode_file / encode_each_byte,
 *      include/canonical_huff_encoder.cc:245-285 + Buffer::write_bits/write_bit/flush_bits,
 *      utils/include/buffer.h:241-248,277-280,290-295 -- as a two-pass scheme.
 * plan: per-chunk bit totals + exclusive scan; *d_total_bits = sum of code lengths of d_in[0..n)
 *       (the end mark is not included).
 * emit: every wave packs one chunk MSB-first into the pre-sized output.
 *   d_start_bit : absolute stream bit of this buffer's first code (device u64); NULL = right after the
 *                 header, i.e. 8*(1040+8*max_len)  (single-GPU case).
 *   flags       : GHF_EMIT_HEADER the header goes out with the same launches (saves the separate ghf_write_header),
 *                 GHF_EMIT_LAST   append the end mark (symbol 256) and pad with 1-bits to a byte,
 *                 GHF_EMIT_REBASE d_out[0] is stream byte 16*(start_bit/128) instead of stream byte 0
 *                                 (shard-local output buffers of the multi-GPU path).
 *   d_index     : optional side-car for ghf_decode (NULL = none).
 *   d_end       : optional device u64[2] = { absolute end bit of what this call wrote (after padding),
 *                 number of bytes of d_out that are now defined }.
 * Both must be called with the same (d_in, n, d_code); plan first. */
#define GHF_EMIT_LAST 1
#define GHF_EMIT_REBASE 2
#define GHF_EMIT_HEADER 4 /* also write the .crs2 header at d_out[0..) (same bytes as ghf_write_header; not with REBASE) */
#define GHF_EMIT_LONG_CODES 8 /* the tables may hold codes of 33..64 bits (ghf_crs_build_code on a tree deeper than 32:
                                 include/huff_tree.cc:157-170 keeps codes as strings of any length): a second, slower kernel
                                 is queued behind the packer and does the work when the first one finds such a code.  The
                                 canonical format never has them (include/canonical_huff_encoder.h:43-44). */
int ghf_encode_plan(ghf_ctx* ctx, const uint8_t* d_in, size_t n, const ghf_code* d_code, uint64_t* d_total_bits);
int ghf_encode_emit(ghf_ctx* ctx, const uint8_t* d_in, size_t n, const ghf_code* d_code, const uint64_t* d_start_bit,
                    int flags, uint8_t* d_out, size_t cap, const ghf_index* index, uint64_t* d_end);

/* Multi-GPU glue (no reference counterpart: the reference is single-stream): given the all-gathered per-rank body
 * bit totals d_totals[world] (device), *d_start_bit = 8*(1040+8*max_len) + sum of d_totals[0..rank) -- the absolute
 * stream bit at which rank `rank` must start emitting.  One tiny kernel, no host synchronisation. */
int ghf_shard_start_bit(ghf_ctx* ctx, const ghf_code* d_code, const uint64_t* d_totals, int world, int rank,
                        uint64_t* d_start_bit);

/* ---- Compressor<CanonicalHuffEncoder<>>::compress(), include/compressor.h:62-73, in one call:
 *      histogram -> code -> header -> plan -> emit, no host synchronisation in between.
 *      d_out receives the complete .crs2 image; d_out_bytes (device u64) its size. */
int ghf_compress(ghf_ctx* ctx, const uint8_t* d_in, size_t n, uint8_t* d_out, size_t cap, uint64_t* d_out_bytes,
                 ghf_code* d_code /* optional out */, const ghf_index* index /* optional */);
size_t ghf_compress_bound(size_t n); /* capacity that always suffices (Huffman never beats 9 bits/symbol on 257 symbols) */

/* ---- side-car index ------------------------------------------------------------------------------ */
uint32_t ghf_chunk_symbols(size_t n); /* the chunk size the library uses for n input bytes */
int ghf_index_alloc(ghf_ctx* ctx, size_t n_symbols, ghf_index* out);
int ghf_index_free(ghf_ctx* ctx, ghf_index* idx);

/* ---- a7: CanonicalHuffDecoder::get_encode_info, include/canonical_huff_encoder.cc:349-374 (host) -
 * Parses and validates a .crs2 header (the reference trusts it blindly). code->length/codeword are
 * reconstructed from symbol/start_pos/first_code. */
int ghf_parse_header(const uint8_t* h_stream, size_t n, ghf_code* code, size_t* header_bytes);

/* ---- K6/K7: decode_file of CanonicalHuffDecoder / FastCanonicalHuffDecoder /
 *      TableCanonicalHuffDecoder, include/canonical_huff_encoder.cc:377-419,422-461,466-568 ---------
 * d_stream is the buffer an emit call wrote (a whole .crs2 image, or one shard's GHF_EMIT_REBASE buffer)
 * and index the side-car that call filled: the decode is then block-parallel (length-indexed canonical
 * table in LDS).  Without one (index == NULL, e.g. a .crs2 written by the reference; d_stream must then
 * be a whole .crs2 image) the index is first rebuilt on the GPU from the bit stream.  stream_bytes may exceed the
 * stream: the first end mark ends it, as in the reference's decoders (canonical_huff_encoder.cc:404), and what lies
 * behind it is neither decoded nor waited for.
 * d_code may come from anywhere: before anything is decoded the tables are checked on the device to be a
 * complete prefix code (Kraft equality, lengths within [min_len, max_len], first codes that fit their length,
 * start positions inside symbol[]; the one-symbol code of GHF_EMPTY_OK is the exception) -- if not,
 * GHF_E_FORMAT is latched and nothing is written.  Every 64-symbol segment's end is checked against the index
 * (GHF_E_CORRUPT).  A stream whose first code is the end mark decodes to nothing without launching K7.
 * d_out_bytes: device u64 = decoded size. */
int ghf_decode(ghf_ctx* ctx, const uint8_t* d_stream, size_t stream_bytes, const ghf_code* d_code,
               const ghf_index* index, uint8_t* d_out, size_t cap, uint64_t* d_out_bytes);

/* Optional: build the decode tables of d_code ahead of time (one small kernel, k_build_decode_tables) so that the
 * next ghf_decode(..., d_code, index != NULL, ...) on this context starts with the decode kernel itself -- e.g. on
 * another stream while the previous buffer is still being emitted.  Single use: the prepared state is consumed by
 * that ghf_decode and dropped by any other call that rebuilds tables on this context.  The caller orders the two
 * streams (event) and must not change *d_code in between. */
int ghf_decode_prepare(ghf_ctx* ctx, const ghf_code* d_code);

/* Size of what a side-car-less stream decodes to (the .crs2 format does not store it: the reference's
 * decoders simply run until the end mark, include/canonical_huff_encoder.cc:404-411).  Rebuilds the
 * side-car on the GPU, synchronises, and keeps it for a following ghf_decode(index = NULL) of the same
 * (d_stream, stream_bytes), which then does not repeat the work. */
int ghf_decoded_size(ghf_ctx* ctx, const uint8_t* d_stream, size_t stream_bytes, const ghf_code* d_code, uint64_t* n_out);

/* Multi-GPU decode of a stream that has no side-car ("per-rank self-sync + one all-gather of symbol
 * counts"; the reference's decoders, canonical_huff_encoder.cc:377-568, are single-stream).  The caller cuts the body
 * at byte positions; a rank's piece is its own bytes followed by >= 8 bytes of look-ahead from the next piece (zeros
 * behind the stream's end).  d_piece / piece_bytes: the piece with its look-ahead; first_bit (< 512): where the first
 * code boundary of the piece is assumed to be; end_bit = 8 * (own bytes): codes that start at or behind it belong to
 * the next piece.  Out (host, the call synchronises): landing = how many bits the last code (or the one in progress)
 * runs past end_bit, i.e. the NEXT piece's first_bit; n_symbols = codes that start in [first_bit, end_bit), up to the
 * end mark if the piece holds it (has_end_mark).  Because Huffman codes self-synchronise, a wrong first_bit only
 * spoils the first few symbols; iterate first_bit[g+1] = landing[g] until nothing changes (sharded.py does), then
 * ghf_decode(d_piece, piece_bytes, d_code, index = NULL, ...) decodes the piece with the side-car this call rebuilt. */
int ghf_sync_piece(ghf_ctx* ctx, const uint8_t* d_piece, size_t piece_bytes, uint32_t first_bit, uint64_t end_bit,
                   const ghf_code* d_code, uint64_t* landing, uint64_t* n_symbols, int* has_end_mark);

/* ------------------------------------------------------------------------------------------------
 * one stream sharded over the GPUs of a node, one process (or thread) and one ghf_ctx per GPU,
 * RCCL over xGMI underneath.  The reference has no counterpart (it is single-threaded, single-stream:
 * include/compressor.h:62-73); what makes the shards ONE .crs2 stream, bit-exact with the single-stream
 * reference, is a single global code, which costs exactly two latency-bound exchanges:
 *   all-reduce(sum) of the 256 byte counts   (the end-mark slot [256] stays 1: it must not be summed)
 *   all-gather of the per-rank body bit totals -> every rank's absolute start bit
 * ghf_comm wraps an ncclComm_t; the RCCL library is bound at first use (dlopen of the copy the process
 * already has, e.g. torch's, else librccl.so.1), so single-GPU users need no RCCL at all.
 * ------------------------------------------------------------------------------------------------ */
typedef struct ghf_comm ghf_comm;
#define GHF_COMM_ID_BYTES 128
int ghf_comm_unique_id(uint8_t id[GHF_COMM_ID_BYTES]);  /* ncclGetUniqueId: one rank calls it, every rank gets the bytes */
int ghf_comm_init_rank(ghf_ctx* ctx, const uint8_t id[GHF_COMM_ID_BYTES], int world, int rank, ghf_comm** out);
int ghf_comm_destroy(ghf_comm* comm);
int ghf_comm_world(const ghf_comm* comm, int* world, int* rank);
int ghf_rccl_version(int* version);                     /* ncclGetVersion of the bound library */
/* the two exchanges, on the context's stream (comm == NULL or world 1: d_totals[0] <- *d_total, nothing else) */
int ghf_comm_allreduce_hist(ghf_ctx* ctx, ghf_comm* comm, uint64_t* d_hist /* [257], in place */);
int ghf_comm_allgather_total(ghf_ctx* ctx, ghf_comm* comm, const uint64_t* d_total, uint64_t* d_totals /* [world] */);
/* This rank's shard of the stream, all of the above in order on the context's stream, no host synchronisation:
 * K1, all-reduce, K2/K3, K4, all-gather, start bit, K5.  rank 0 writes the header and d_out[0] is stream byte 0;
 * rank g > 0 writes from stream byte 16 * (start_bit / 128) on (GHF_EMIT_REBASE); the last rank appends the end mark.
 * cap >= ghf_shard_bound(n): a shard is packed with the GLOBAL code, which may be far from optimal for it.
 * d_code <- the (identical on every rank) tables; *d_start_bit <- this shard's absolute first bit;
 * d_end[0..1] <- {absolute end bit, bytes defined in d_out}; index (optional) <- side-car for ghf_decode of the shard
 * (index->flags is set: GHF_INDEX_NO_END_MARK on every rank but the last). */
int ghf_encode_sharded(ghf_ctx* ctx, ghf_comm* comm, const uint8_t* d_in, size_t n, uint8_t* d_out, size_t cap,
                       ghf_code* d_code, ghf_index* index, uint64_t* d_start_bit, uint64_t* d_end);
size_t ghf_shard_bound(size_t n); /* header + 4 n (32 bits per symbol) + end mark + alignment slack: what fits ANY code */
/* The EXACT number of bytes this rank's K5 will define in d_out, from the all-gathered bit totals (ghf_comm_allgather_total)
 * and the global code -- both known before K5 runs.  A caller that can afford ONE host synchronisation per stream (the first
 * step of a pipeline, or a caller that is not pipelined) sizes its shard outputs with this instead of ghf_shard_bound: at
 * BASELINE config 4 (4 GiB of uniform bytes per rank) that is 4.3 GB per buffer instead of 17.2.  Waits for the context's
 * stream.  *bytes = what ghf_encode_emit / ghf_encode_sharded need as `cap` (whole 16-byte units, the one a shard shares with
 * its right neighbour included); the same number K5 reports in d_end[1] afterwards, rounded up to its last unit. */
int ghf_shard_bytes(ghf_ctx* ctx, const ghf_code* d_code, const uint64_t* d_totals, int world, int rank, size_t* bytes);

/* ------------------------------------------------------------------------------------------------
 * inputs on which the reference is undefined because a code would be longer than 32 bits
 * (include/canonical_huff_encoder.h:43-44; needs > 14.9 M bytes with Fibonacci-like counts).  With GHF_CODE_LIMIT
 * the code lengths are then replaced by the OPTIMAL lengths under a 32-bit limit (package-merge; leaves ordered by
 * (count ascending, byte value ascending), a leaf before a package of equal weight), and the usual canonical
 * assignment (include/canonical_huff_encoder.cc:69-141) follows: the result is an ordinary .crs2 that the
 * reference's decoders read.  Whenever the reference-exact code fits 32 bits the flag changes nothing, so the
 * default output stays bit-exact.
 * ------------------------------------------------------------------------------------------------ */
#define GHF_CODE_LIMIT 1u
/* Opt-in, second half of N4: the EMPTY input.  The reference is undefined there (its merge loop, include/
 * canonical_huff_encoder.cc:309-343, runs n - 1 = 0 times over the lone end mark and leaves every length 0).  With
 * GHF_EMPTY_OK, ghf_build_code_ex gives the end mark the one-bit code "0" (min_len = max_len = 1) and ghf_compress_ex
 * (n == 0) writes the 1048-byte header of that code followed by the byte 0x7F (the end mark, then 1-bits up to the
 * byte, as flush_bits pads): 1049 bytes that ghf_parse_header accepts and ghf_decode turns back into nothing.
 * This is the builder's own definition -- PARITY UNPINNED: no reference output exists to compare with.  The staged
 * calls (ghf_encode_plan / ghf_encode_emit) keep refusing n == 0. */
#define GHF_EMPTY_OK 2u
int ghf_build_code_ex(ghf_ctx* ctx, const uint64_t* d_hist, ghf_code* d_code, unsigned flags);
int ghf_compress_ex(ghf_ctx* ctx, const uint8_t* d_in, size_t n, uint8_t* d_out, size_t cap, uint64_t* d_out_bytes,
                    ghf_code* d_code, const ghf_index* index, unsigned code_flags);

/* ------------------------------------------------------------------------------------------------
 * the `.crs` format -- Compressor<NormalHuffEncoder<>> / Decompressor<NormalHuffDecoder<>>
 * (include/normal_huff_encoder.h, include/huff_tree.h, include/huff_tree.cc).  Same histogram (256 byte values,
 * no end mark), the Huffman TREE itself defines the codes ('0' = left = first popped, '1' = right), the file is
 *   [tree in preorder, 2 bytes per node: (0, key) leaf / (255, 255) parent] [left_bits] [last byte] [whole body bytes]
 * The kernels are the ones above (K1, K4, K5, K7, K6); only the code assignment and the framing differ.
 * The reference keeps codes as strings of any length (include/huff_tree.cc:157-170): a tree deeper than 32 -- more than
 * 3.5 M input bytes with counts arranged like Fibonacci numbers -- is packed by a second, slower kernel
 * (GHF_EMIT_LONG_CODES) and decoded by a 64-bit tree walk.  Depths beyond 64 (> 2^44 input bytes) are refused
 * (GHF_E_CODELEN).
 * ------------------------------------------------------------------------------------------------ */

/* The tree as NormalHuffEncoder builds it (EncodeHuffTree, include/huff_tree.h:175-262) and NormalHuffDecoder
 * rebuilds it (DecodeHuffTree::do_build_tree, include/huff_tree.cc:289-303).  Node ids: 0..255 = leaf with that
 * key, 256 + i = the i-th parent; left[i] / right[i] are the children of parent i. */
typedef struct ghf_tree {
  uint16_t left[256];
  uint16_t right[256];
  uint32_t root;        /* node id; >= 256 (a tree that is a single leaf is refused) */
  uint32_t n_leaves;    /* 2 .. 256 */
  uint32_t max_len;     /* depth of the deepest leaf, <= 64 */
  uint32_t tree_bytes;  /* 2 * (2 * n_leaves - 1) */
  uint8_t header[1024]; /* the preorder serialisation (tree_bytes of it), huff_tree.cc:174-187 */
} ghf_tree;

/* EncodeHuffTree::build_tree + gen_encode + serialize_tree (include/huff_tree.cc:138-187) on one wavefront.
 * d_hist: ghf_histogram()'s output (slot [256] is ignored).  d_code receives length[] / codeword[] for K4/K5; when the
 * tree is deeper than 32, codeword[] holds bits 0..31 of every code and symbol[] bits 32..63 (else symbol[] = 0xFFFFFFFF). */
int ghf_crs_build_code(ghf_ctx* ctx, const uint64_t* d_hist, ghf_tree* d_tree, ghf_code* d_code);

/* Compressor<NormalHuffEncoder<>>::compress() (include/compressor.h:62-73, normal_huff_encoder.h:136-186).
 * d_out_bytes <- size of the .crs image.  If the body ends inside a byte, that byte (zero-filled) is ALSO left in
 * d_out right behind the image, which is the layout ghf_crs_decode() wants.  d_tree (optional) receives the tree. */
int ghf_crs_compress(ghf_ctx* ctx, const uint8_t* d_in, size_t n, uint8_t* d_out, size_t cap, uint64_t* d_out_bytes,
                     ghf_tree* d_tree, const ghf_index* index);
size_t ghf_crs_compress_bound(size_t n);

/* DecodeHuffTree::build_tree (include/huff_tree.cc:289-303) on the host, with the checks the reference lacks
 * (truncated / over-long / degenerate trees -> GHF_E_FORMAT, depth > 64 -> GHF_E_CODELEN).  *tree_bytes <- size of
 * the serialised tree; the two prefix bytes {left_bits, last byte} follow it (normal_huff_encoder.h:163-164). */
int ghf_crs_parse_header(const uint8_t* h_stream, size_t n, ghf_tree* tree, size_t* tree_bytes);

/* Decompressor<NormalHuffDecoder<>>::decompress() (include/huff_tree.cc:191-207,255-271).
 * d_stream / stream_bytes: the .crs image from its first byte, with -- when left_bits != 0 -- the stored last byte
 * appended behind the body (so that the code bits are contiguous); stream_bytes counts that byte.
 * index == NULL: the side-car is rebuilt on the GPU (K6) and the stream must end exactly on a code boundary,
 * left_bits bits before its last byte ends. */
int ghf_crs_decode(ghf_ctx* ctx, const uint8_t* d_stream, size_t stream_bytes, int left_bits, const ghf_tree* d_tree,
                   const ghf_index* index, uint8_t* d_out, size_t cap, uint64_t* d_out_bytes);
int ghf_crs_decoded_size(ghf_ctx* ctx, const uint8_t* d_stream, size_t stream_bytes, int left_bits, const ghf_tree* d_tree,
                         uint64_t* n_out);
/* A .crs body in pieces (the file pipeline of the host layer; the format has no sync points and no end mark either):
 * as ghf_sync_piece, with the tree instead of canonical tables.  d_piece: the piece's own bytes + >= 8 bytes of
 * look-ahead (for the last piece: the stored last byte, then zeros); end_bit = 8 * own bytes, for the last piece minus
 * left_bits of the stored byte that was appended.  landing = where the next piece's first code begins (the last piece
 * must land on 0: the stream ends on a code boundary); n_symbols = codes that start in [first_bit, end_bit).
 * ghf_crs_decode(d_piece, piece_bytes, 0, d_tree, NULL, ...) then decodes the piece with the side-car this call rebuilt. */
int ghf_crs_sync_piece(ghf_ctx* ctx, const uint8_t* d_piece, size_t piece_bytes, uint32_t first_bit, uint64_t end_bit,
                       const ghf_tree* d_tree, uint64_t* landing, uint64_t* n_symbols);

#ifdef __cplusplus
}
#endif
#endif /* GHF_H_ */
