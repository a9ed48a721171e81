/* oracle/huff_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's byte-keyed canonical-Huffman path
 * (chenghuige/golden-huffman, `glzip`).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may use it -- as the checker, never as the product.
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference).  Parity is PINNED: tests/test_oracle_golden.py checks it against the
 * fixtures under tests/golden/, which were produced by the compiled reference itself
 * (oracle/_ref/ref_glzip, recipe in oracle/Makefile, generator tests/golden/make_golden.py).
 */
#ifndef HUFF_ORACLE_H_
#define HUFF_ORACLE_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NSYM 257 /* include/type_traits.h:50  CharSymbolNum: 256 bytes + end-of-stream mark */

/* mirrors the private tables of CanonicalHuffEncoder, include/canonical_huff_encoder.h:107-120 */
typedef struct orc_code {
  uint32_t length[ORC_NSYM];
  uint32_t codeword[ORC_NSYM];
  uint32_t symbol[ORC_NSYM];
  uint32_t first_code[64];
  uint32_t start_pos[64];
  int32_t min_len;
  int32_t max_len;
} orc_code;

/* include/encoder.h:123-129 (init, freq[256]=1) + :136-150 (byte loop). */
void orc_histogram(const uint8_t* in, size_t n, int64_t hist[ORC_NSYM]);

/* include/canonical_huff_encoder.cc:289-345 with the std::priority_queue of
 * include/canonical_huff_encoder.h:58-70 restated as libstdc++'s push_heap/pop_heap.
 * `hist` is mutated exactly like frequency_map_ is (survivor accumulates). Returns max_len. */
int orc_code_lengths(int64_t hist[ORC_NSYM], uint32_t length[ORC_NSYM]);

/* include/canonical_huff_encoder.cc:69-141 (do_gen_encode). length[] and max_len must be set. */
void orc_canonical(orc_code* c);

/* histogram -> lengths -> canonical tables. returns 0, or -1 if n==0 (reference undefined),
 * -2 if max_len>32 (reference limit, include/canonical_huff_encoder.h:43-44). */
int orc_build_code(const int64_t hist_in[ORC_NSYM], orc_code* c);

/* include/canonical_huff_encoder.cc:210-242 + utils/include/buffer.h:255-268. returns header bytes. */
size_t orc_header_size(const orc_code* c);
size_t orc_write_header(const orc_code* c, uint8_t* out);

/* include/canonical_huff_encoder.cc:245-285 + buffer.h:241-248,277-280,290-295: per-bit MSB-first
 * packer, EOF code, 1-padding. returns body bytes written (or (size_t)-1 when cap is too small). */
size_t orc_encode_body(const uint8_t* in, size_t n, const orc_code* c, uint8_t* out, size_t cap);

/* Compressor<CanonicalHuffEncoder<>>::compress(), include/compressor.h:62-73. returns 0 ok. */
int orc_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_n);
size_t orc_compress_bound(size_t n);

/* include/canonical_huff_encoder.cc:349-374 (get_encode_info). returns header bytes consumed, 0 on error. */
size_t orc_parse_header(const uint8_t* in, size_t n, orc_code* c);

/* include/canonical_huff_encoder.cc:377-419: bit-serial decoder, stops on symbol 256.
 * returns 0 ok, -1 input exhausted before the end mark, -2 output cap too small. */
int orc_decode_body(const uint8_t* body, size_t body_n, const orc_code* c, uint8_t* out, size_t cap, size_t* out_n);
int orc_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_n);

/* sharded-path helper (tests only): pack at a bit phase, optionally with end mark + padding */
uint64_t orc_pack_at(const uint8_t* in, size_t n, const orc_code* c, unsigned phase, int last, uint8_t* out, size_t cap);

/* total body bits incl. the EOF code, before padding */
uint64_t orc_body_bits(const int64_t hist[ORC_NSYM], const orc_code* c);

/* ---------------------------------------------------------------- the .crs format
 * (include/normal_huff_encoder.h, include/huff_tree.h, include/huff_tree.cc) */
#define ORC_CRS_NODES 512 /* 256 leaves (node id = key) + up to 255 parents (256 + creation order) */
typedef struct orc_tree {
  int left[ORC_CRS_NODES], right[ORC_CRS_NODES]; /* -1 at leaves */
  int root, n_leaves, n_nodes;
} orc_tree;
typedef struct orc_crs_code {
  uint16_t len[256];
  uint8_t bits[256][256]; /* one 0/1 per code bit, root first (the reference keeps std::string codes) */
} orc_crs_code;

int orc_crs_build_tree(const int64_t hist256[256], orc_tree* t);
void orc_crs_codes(const orc_tree* t, orc_crs_code* c);
size_t orc_crs_write_tree(const orc_tree* t, uint8_t* out);
size_t orc_crs_parse_tree(const uint8_t* in, size_t n, orc_tree* t);
size_t orc_crs_bound(size_t n);
int orc_crs_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_n);
int orc_crs_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_n);

/* ---------------------------------------------------------------- opt-in length limit (not reference behaviour) */
int orc_limit_lengths(const int64_t hist[ORC_NSYM], uint32_t length[ORC_NSYM], int limit);
int orc_build_code_limited(const int64_t hist_in[ORC_NSYM], orc_code* c, int limit);
int orc_compress_limited(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_n, int limit);

#ifdef __cplusplus
}
#endif
#endif
