/* oracle/huff_oracle.c -- TEST INFRASTRUCTURE ONLY (see huff_oracle.h).
 *
 * CPU restatement of the reference algorithm for the byte-keyed canonical-Huffman path.
 * Written from the behaviour of the reference (file:line cited per function, relative to
 * /root/reference); no reference source is reproduced here.
 *
 * Third-party arithmetic on the path: libstdc++'s std::priority_queue / push_heap / pop_heap
 * (GCC 11.4 <bits/stl_heap.h>, not vendored by the reference).  Its published algorithm
 * (__push_heap: sift the hole up while comp(parent, value); __adjust_heap: walk the hole to
 * the bottom always taking the child for which comp(right, left) is false -> right, else left,
 * handle the lone left child of an even-length heap, then __push_heap the displaced value) is
 * restated in heap_push()/heap_pop() below; fixtures tests/golden/ pin it.
 */
#include "huff_oracle.h"
#include <string.h>

/* ------------------------------------------------------------------ a1/a2: histogram */
/* include/encoder.h:123-129: freq[0..255]=0, freq[256]=1; :136-150: freq[byte]++ */
void orc_histogram(const uint8_t* in, size_t n, int64_t hist[ORC_NSYM]) {
  for (int i = 0; i < ORC_NSYM - 1; i++) hist[i] = 0;
  hist[ORC_NSYM - 1] = 1;
  /* the reference reads 64 KiB blocks (include/encoder.h:55,143); chunking does not change counts */
  for (size_t i = 0; i < n; i++) hist[in[i]] += 1;
}

/* ------------------------------------------------------------------ a3: code lengths */
/* comparator of include/canonical_huff_encoder.h:58-66: comp(a,b) = freq[a] > freq[b] (min-heap on freq,
 * reading the *live* frequency table) */
typedef struct {
  int a[ORC_NSYM];
  int n;
  const int64_t* f;
} orc_heap;

static int cmp_gt(const orc_heap* h, int x, int y) { return h->f[x] > h->f[y]; }

/* libstdc++ __push_heap(first, holeIndex, topIndex, value, comp) */
static void sift_up(orc_heap* h, int hole, int top, int value) {
  int parent = (hole - 1) / 2;
  while (hole > top && cmp_gt(h, h->a[parent], value)) {
    h->a[hole] = h->a[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  h->a[hole] = value;
}

/* priority_queue::push = push_back + push_heap */
static void heap_push(orc_heap* h, int v) {
  h->a[h->n] = v;
  h->n++;
  sift_up(h, h->n - 1, 0, v);
}

/* priority_queue::pop = pop_heap + pop_back; pop_heap moves a[0] to the back, then
 * __adjust_heap(first, 0, len = n-1, value = old back) */
static void heap_pop(orc_heap* h) {
  if (h->n > 1) {
    int len = h->n - 1;
    int value = h->a[len];
    h->a[len] = h->a[0];
    int hole = 0, child = 0;
    while (child < (len - 1) / 2) {
      child = 2 * (child + 1);
      if (cmp_gt(h, h->a[child], h->a[child - 1])) child--;
      h->a[hole] = h->a[child];
      hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
      child = 2 * (child + 1);
      h->a[hole] = h->a[child - 1];
      hole = child - 1;
    }
    sift_up(h, hole, 0, value);
  }
  h->n--;
}

/* include/canonical_huff_encoder.cc:289-345 (get_encoding_length) */
int orc_code_lengths(int64_t hist[ORC_NSYM], uint32_t length[ORC_NSYM]) {
  int group[ORC_NSYM];
  orc_heap h;
  h.n = 0;
  h.f = hist;
  for (int i = 0; i < ORC_NSYM; i++) { /* .cc:301-306 */
    if (hist[i]) heap_push(&h, i);
    group[i] = -1;
    length[i] = 0;
  }
  int times = h.n - 1; /* .cc:309 */
  for (int t = 0; t < times; t++) {
    int top1 = h.a[0];
    heap_pop(&h);
    int top2 = h.a[0];
    heap_pop(&h);
    int index = top2; /* .cc:316-329: +1 for every member of both chains, chain1 appended to chain2 */
    while (group[index] != -1) {
      length[index] += 1;
      index = group[index];
    }
    group[index] = top1;
    while (index != -1) {
      length[index] += 1;
      index = group[index];
    }
    hist[top2] += hist[top1]; /* .cc:331 survivor = second popped */
    heap_push(&h, top2);      /* .cc:333 */
  }
  uint32_t mx = 0; /* .cc:343 */
  for (int i = 0; i < ORC_NSYM; i++)
    if (length[i] > mx) mx = length[i];
  return (int)mx;
}

/* ------------------------------------------------------------------ a4: canonical assignment */
/* include/canonical_huff_encoder.cc:69-141 */
void orc_canonical(orc_code* c) {
  uint32_t num[66], next_code[66], spos[66];
  int max_len = c->max_len;
  memset(num, 0, sizeof num);
  for (int i = 0; i < 64; i++) {
    c->first_code[i] = 0;
    c->start_pos[i] = 0;
  }
  for (int i = 0; i < ORC_NSYM; i++) { /* .cc:85-89 */
    num[c->length[i]] += 1;
    c->symbol[i] = 0xFFFFFFFFu;
    c->codeword[i] = 0;
  }
  num[0] = 0;
  c->min_len = 0;
  for (int i = 1; i <= max_len; i++) /* .cc:93-98 */
    if (num[i]) {
      c->min_len = i;
      break;
    }
  for (int i = 1; i <= max_len; i++) c->start_pos[i] = num[i - 1] + c->start_pos[i - 1]; /* .cc:104-105 */
  c->first_code[max_len] = 0; /* .cc:109-114 */
  next_code[max_len] = 0;
  for (int i = max_len - 1; i >= 1; i--) {
    c->first_code[i] = (c->first_code[i + 1] + num[i + 1]) / 2;
    next_code[i] = c->first_code[i];
  }
  for (int i = 1; i < c->min_len; i++) c->first_code[i] = 1024; /* .cc:119-121 sentinel */
  for (int i = 0; i <= max_len; i++) spos[i] = c->start_pos[i];
  for (int i = 0; i < ORC_NSYM; i++) { /* .cc:127-133 */
    uint32_t len = c->length[i];
    if (len) {
      c->codeword[i] = next_code[len]++;
      c->symbol[spos[len]++] = (uint32_t)i;
    }
  }
}

int orc_build_code(const int64_t hist_in[ORC_NSYM], orc_code* c) {
  int64_t hist[ORC_NSYM];
  int nz = 0;
  memcpy(hist, hist_in, sizeof hist);
  for (int i = 0; i < 256; i++) nz += hist[i] != 0;
  if (nz == 0) return -1; /* empty input: reference undefined */
  memset(c, 0, sizeof *c);
  c->max_len = orc_code_lengths(hist, c->length);
  if (c->max_len > 32) return -2;
  orc_canonical(c);
  return 0;
}

/* ------------------------------------------------------------------ a5: header */
static uint8_t* put_be32(uint8_t* p, uint32_t v) { /* utils/include/buffer.h:261-268 big-endian */
  p[0] = (uint8_t)(v >> 24);
  p[1] = (uint8_t)(v >> 16);
  p[2] = (uint8_t)(v >> 8);
  p[3] = (uint8_t)v;
  return p + 4;
}
static uint32_t get_be32(const uint8_t* p) { /* buffer.h:194-206 */
  return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

size_t orc_header_size(const orc_code* c) { return 4 + 4 * ORC_NSYM + 8 + 8 * (size_t)c->max_len; }

/* include/canonical_huff_encoder.cc:210-242 */
size_t orc_write_header(const orc_code* c, uint8_t* out) {
  uint8_t* p = out;
  p = put_be32(p, ORC_NSYM);
  for (int i = 0; i < ORC_NSYM; i++) p = put_be32(p, c->symbol[i]);
  p = put_be32(p, (uint32_t)c->min_len);
  p = put_be32(p, (uint32_t)c->max_len);
  for (int i = 1; i <= c->max_len; i++) {
    p = put_be32(p, c->start_pos[i]);
    p = put_be32(p, c->first_code[i]);
  }
  return (size_t)(p - out);
}

/* ------------------------------------------------------------------ a6: body */
typedef struct {
  uint8_t* out;
  size_t cap, cur;
  unsigned num;
  int bit_cur;
  int overflow;
} orc_bitw;

/* buffer.h:241-248 write_bit (+ :234-238 write_byte) */
static void w_bit(orc_bitw* w, int x) {
  w->num = (w->num << 1) | (unsigned)x;
  if (++w->bit_cur == 8) {
    if (w->cur < w->cap)
      w->out[w->cur++] = (uint8_t)w->num;
    else
      w->overflow = 1;
    w->num = 0;
    w->bit_cur = 0;
  }
}
/* buffer.h:290-295 write_bits: MSB first */
static void w_bits(orc_bitw* w, uint32_t code, int nbits) {
  for (int i = nbits - 1; i >= 0; i--) w_bit(w, (int)((code >> i) & 1u));
}

/* include/canonical_huff_encoder.cc:245-285 */
size_t orc_encode_body(const uint8_t* in, size_t n, const orc_code* c, uint8_t* out, size_t cap) {
  orc_bitw w = {out, cap, 0, 0, 0, 0};
  for (size_t i = 0; i < n; i++) w_bits(&w, c->codeword[in[i]], (int)c->length[in[i]]);
  w_bits(&w, c->codeword[ORC_NSYM - 1], (int)c->length[ORC_NSYM - 1]); /* .cc:255 end mark */
  while ((8 - w.bit_cur) % 8) w_bit(&w, 1);                             /* buffer.h:272-280 flush_bits pads 1s */
  return w.overflow ? (size_t)-1 : w.cur;
}

/* test helper for the sharded (multi-GPU) path, SURVEY 8(e): pack the codes of in[0..n) MSB-first starting at bit
 * `phase` (0..7) of out[0] -- the bits in front stay zero --, optionally followed by the end mark and the
 * 1-padding (canonical_huff_encoder.cc:255-257).  Returns the bits written behind the phase, or (uint64_t)-1. */
uint64_t orc_pack_at(const uint8_t* in, size_t n, const orc_code* c, unsigned phase, int last, uint8_t* out, size_t cap) {
  orc_bitw w = {out, cap, 0, 0, 0, 0};
  uint64_t bits = 0;
  for (unsigned i = 0; i < phase; i++) w_bit(&w, 0);
  for (size_t i = 0; i < n; i++) {
    w_bits(&w, c->codeword[in[i]], (int)c->length[in[i]]);
    bits += c->length[in[i]];
  }
  if (last) {
    w_bits(&w, c->codeword[ORC_NSYM - 1], (int)c->length[ORC_NSYM - 1]);
    bits += c->length[ORC_NSYM - 1];
    while ((8 - w.bit_cur) % 8) {
      w_bit(&w, 1);
      bits++;
    }
  } else {
    while ((8 - w.bit_cur) % 8) w_bit(&w, 0); /* the next shard ORs its bits in here */
  }
  return w.overflow ? (uint64_t)-1 : bits;
}

uint64_t orc_body_bits(const int64_t hist[ORC_NSYM], const orc_code* c) {
  uint64_t bits = 0;
  for (int i = 0; i < 256; i++) bits += (uint64_t)hist[i] * c->length[i];
  return bits + c->length[256];
}

size_t orc_compress_bound(size_t n) { return 4 + 4 * ORC_NSYM + 8 + 8 * 32 + (n * 9 + 9 + 7) / 8 + 16; }

/* include/compressor.h:62-73 */
int orc_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_n) {
  int64_t hist[ORC_NSYM];
  orc_code c;
  orc_histogram(in, n, hist);
  int rc = orc_build_code(hist, &c);
  if (rc) return rc;
  size_t hs = orc_header_size(&c);
  if (cap < hs) return -3;
  orc_write_header(&c, out);
  size_t bs = orc_encode_body(in, n, &c, out + hs, cap - hs);
  if (bs == (size_t)-1) return -3;
  *out_n = hs + bs;
  return 0;
}

/* ------------------------------------------------------------------ a7: header parse */
/* include/canonical_huff_encoder.cc:349-374 */
size_t orc_parse_header(const uint8_t* in, size_t n, orc_code* c) {
  memset(c, 0, sizeof *c);
  if (n < 4 + 4 * ORC_NSYM + 8) return 0;
  uint32_t nsym = get_be32(in);
  if (nsym != ORC_NSYM) return 0;
  const uint8_t* p = in + 4;
  for (int i = 0; i < ORC_NSYM; i++, p += 4) c->symbol[i] = get_be32(p);
  c->min_len = (int32_t)get_be32(p);
  p += 4;
  c->max_len = (int32_t)get_be32(p);
  p += 4;
  if (c->max_len < 1 || c->max_len > 32 || c->min_len < 1 || c->min_len > c->max_len) return 0;
  if (n < (size_t)(p - in) + 8 * (size_t)c->max_len) return 0;
  for (int i = 1; i <= c->max_len; i++) {
    c->start_pos[i] = get_be32(p);
    p += 4;
    c->first_code[i] = get_be32(p);
    p += 4;
  }
  return (size_t)(p - in);
}

/* ------------------------------------------------------------------ a8: bit-serial decode */
/* include/canonical_huff_encoder.cc:377-419 */
int orc_decode_body(const uint8_t* body, size_t body_n, const orc_code* c, uint8_t* out, size_t cap, size_t* out_n) {
  uint32_t v = 0;
  int len = 0;
  size_t o = 0;
  for (size_t i = 0; i < body_n; i++) {
    uint8_t ch = body[i];
    for (int b = 7; b >= 0; b--) {
      v = (v << 1) | ((ch >> b) & 1u);
      len++;
      if (len > c->max_len) return -1; /* the reference would index past its tables here */
      if (v >= c->first_code[len]) {
        uint32_t idx = c->start_pos[len] + v - c->first_code[len];
        if (idx >= ORC_NSYM) return -1;
        uint32_t sym = c->symbol[idx];
        if (sym == ORC_NSYM - 1) {
          *out_n = o;
          return 0;
        }
        if (o >= cap) return -2;
        out[o++] = (uint8_t)sym;
        v = 0;
        len = 0;
      }
    }
  }
  *out_n = o;
  return -1;
}

int orc_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_n) {
  orc_code c;
  size_t hs = orc_parse_header(in, n, &c);
  if (!hs) return -4;
  return orc_decode_body(in + hs, n - hs, &c, out, cap, out_n);
}

/* ================================================================== SURVEY 8(f) N3: the .crs format
 * (NormalHuffEncoder / NormalHuffDecoder, include/normal_huff_encoder.h + include/huff_tree.{h,cc}).
 * Same test-infrastructure status as everything above; pinned by tests/golden/golden_crs.json, which
 * the compiled reference (ref_glzip nc/nd/nt) produced. */

/* include/huff_tree.h:228-235 (init_queue: keys 0..255 ascending, zero counts skipped; no end mark) and
 * include/huff_tree.cc:138-153 (build_tree: left = first popped, right = second popped, push the parent).
 * The queue is std::priority_queue<Node*, deque<Node*>, HuffNodePtrGreater> (huff_tree.h:190-201): the same
 * libstdc++ heap on weight only as above.  Nodes 0..255 are the leaves (by key), 256+t is the t-th parent. */
int orc_crs_build_tree(const int64_t hist256[256], orc_tree* t) {
  int64_t w[ORC_CRS_NODES];
  orc_heap h;
  h.n = 0;
  h.f = w;
  memset(t, 0, sizeof(*t));
  for (int i = 0; i < ORC_CRS_NODES; i++) t->left[i] = t->right[i] = -1;
  for (int i = 0; i < 256; i++) {
    w[i] = hist256[i];
    if (hist256[i]) {
      heap_push(&h, i);
      t->n_leaves++;
    }
  }
  if (t->n_leaves == 0) return -1; /* empty input: pqueue_.top() on an empty queue, undefined */
  int times = h.n - 1;
  for (int k = 0; k < times; k++) {
    int l = h.a[0];
    heap_pop(&h);
    int r = h.a[0];
    heap_pop(&h);
    int p = 256 + k;
    t->left[p] = l;
    t->right[p] = r;
    w[p] = w[l] + w[r]; /* huff_tree.h:62-66 */
    heap_push(&h, p);
  }
  t->root = h.a[0];
  t->n_nodes = 2 * t->n_leaves - 1;
  return t->n_leaves == 1 ? -3 : 0; /* a lone leaf gets the empty code: the reference's decoder then dereferences NULL */
}

/* include/huff_tree.cc:158-171 (do_gen_encode: '0' to the left, '1' to the right, preorder) */
static void crs_codes_rec(const orc_tree* t, int node, uint8_t* path, int depth, orc_crs_code* c) {
  if (t->left[node] < 0) {
    c->len[node] = (uint16_t)depth;
    memcpy(c->bits[node], path, (size_t)depth);
    return;
  }
  path[depth] = 0;
  crs_codes_rec(t, t->left[node], path, depth + 1, c);
  path[depth] = 1;
  crs_codes_rec(t, t->right[node], path, depth + 1, c);
}

void orc_crs_codes(const orc_tree* t, orc_crs_code* c) {
  uint8_t path[256];
  memset(c, 0, sizeof(*c));
  crs_codes_rec(t, t->root, path, 0, c);
}

/* include/huff_tree.cc:174-187 (do_serialize_tree): preorder, two bytes per node: (0, key) / (255, 255) */
static uint8_t* crs_ser_rec(const orc_tree* t, int node, uint8_t* p) {
  if (t->left[node] < 0) {
    *p++ = 0;
    *p++ = (uint8_t)node;
    return p;
  }
  *p++ = 255;
  *p++ = 255;
  p = crs_ser_rec(t, t->left[node], p);
  return crs_ser_rec(t, t->right[node], p);
}

size_t orc_crs_write_tree(const orc_tree* t, uint8_t* out) { return (size_t)(crs_ser_rec(t, t->root, out) - out); }

size_t orc_crs_bound(size_t n) { return 2 * 511 + 2 + n + n / 4 + 64; } /* byte Huffman averages < 9 bits */

/* Compressor<NormalHuffEncoder<>>::compress(), include/compressor.h:62-73 with
 * include/normal_huff_encoder.h:136-138 (tree first) and :159-186 (two placeholder bytes, every code MSB-first,
 * then -- if the last byte is incomplete -- {left_bits, last byte zero-filled} go back into the placeholders;
 * the body keeps whole bytes only, utils/include/buffer.h:233-247,277-280). returns 0, -1 empty, -3 single symbol */
int orc_crs_compress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_n) {
  int64_t hist[256];
  memset(hist, 0, sizeof(hist));
  for (size_t i = 0; i < n; i++) hist[in[i]]++; /* include/encoder.h:136-150 */
  orc_tree t;
  int rc = orc_crs_build_tree(hist, &t);
  if (rc) return rc;
  static orc_crs_code c; /* 64 KiB: not on the stack */
  orc_crs_codes(&t, &c);
  if (cap < 2 * (size_t)t.n_nodes + 2) return -2;
  size_t hs = orc_crs_write_tree(&t, out);
  uint8_t* body = out + hs + 2;
  size_t nb = 0;
  unsigned acc = 0, k = 0;
  for (size_t i = 0; i < n; i++) {
    const uint8_t* b = c.bits[in[i]];
    for (unsigned j = 0; j < c.len[in[i]]; j++) {
      acc = (acc << 1) | b[j];
      if (++k == 8) {
        if (hs + 2 + nb >= cap) return -2;
        body[nb++] = (uint8_t)acc;
        acc = 0;
        k = 0;
      }
    }
  }
  unsigned left = (8 - k) % 8;
  out[hs] = (uint8_t)left;
  out[hs + 1] = left ? (uint8_t)(acc << left) : 0;
  *out_n = hs + 2 + nb;
  return 0;
}

/* include/huff_tree.cc:289-303 (do_build_tree: preorder, first byte 0 = leaf) -- with the bounds checks the
 * reference lacks.  returns header bytes, 0 on a malformed/truncated tree. */
static int crs_parse_rec(const uint8_t* in, size_t n, size_t* pos, orc_tree* t, int* next_internal, int depth) {
  if (*pos + 2 > n || depth > 256) return -1;
  uint8_t first = in[*pos], second = in[*pos + 1];
  *pos += 2;
  if (first == 0) {
    t->n_leaves++;
    return second;
  }
  if (*next_internal >= ORC_CRS_NODES) return -1;
  int p = (*next_internal)++;
  int l = crs_parse_rec(in, n, pos, t, next_internal, depth + 1);
  if (l < 0) return -1;
  int r = crs_parse_rec(in, n, pos, t, next_internal, depth + 1);
  if (r < 0) return -1;
  t->left[p] = l;
  t->right[p] = r;
  return p;
}

size_t orc_crs_parse_tree(const uint8_t* in, size_t n, orc_tree* t) {
  memset(t, 0, sizeof(*t));
  for (int i = 0; i < ORC_CRS_NODES; i++) t->left[i] = t->right[i] = -1;
  size_t pos = 0;
  int next_internal = 256;
  int root = crs_parse_rec(in, n, &pos, t, &next_internal, 0);
  if (root < 0) return 0;
  t->root = root;
  t->n_nodes = 2 * t->n_leaves - 1;
  return pos;
}

/* Decompressor<NormalHuffDecoder<>>::decompress(): include/huff_tree.cc:191-207 (two prefix bytes, every body byte
 * bit by bit from the MSB, then 8 - left_bit bits of the stored last byte) and :255-271 (decode_byte: walk, emit at
 * a leaf, restart at the root).  returns 0, -1 malformed, -2 cap, -3 root is a leaf (reference: NULL dereference) */
int orc_crs_decompress(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_n) {
  orc_tree t;
  size_t hs = orc_crs_parse_tree(in, n, &t);
  if (!hs || hs + 2 > n) return -1;
  if (t.left[t.root] < 0) return -3;
  unsigned left = in[hs], last = in[hs + 1];
  if (left > 7) return -1;
  size_t no = 0;
  int cur = t.root;
  for (size_t i = hs + 2; i <= n; i++) {
    unsigned byte, nbits;
    if (i < n) {
      byte = in[i];
      nbits = 8;
    } else {
      if (!left) break;
      byte = last;
      nbits = 8 - left;
    }
    for (unsigned b = 0; b < nbits; b++) {
      cur = ((byte >> (7 - b)) & 1) ? t.right[cur] : t.left[cur];
      if (t.left[cur] < 0) {
        if (no >= cap) return -2;
        out[no++] = (uint8_t)cur;
        cur = t.root;
      }
    }
  }
  *out_n = no;
  return 0;
}

/* ================================================================== SURVEY 8(f) N4: length-limited codes (opt-in)
 * NOT a restatement of the reference: include/canonical_huff_encoder.h:43-44 simply cannot represent a code longer
 * than 32 bits, so on such inputs the reference is undefined.  This is the definition libghf.so's opt-in
 * GHF_CODE_LIMIT follows, restated independently so that the GPU result can be checked; what pins it to the
 * reference is (a) inputs whose codes fit are untouched (bit-exact as before) and (b) the reference's own DECODER
 * reads the limited streams back (tests/test_gpu_limit.py, compiled reference).
 *
 * Method: package-merge (Larmore & Hirschberg 1990), the optimal length-limited prefix code.  Leaves sorted by
 * (frequency ascending, index ascending); for each level d = limit .. 1 the list of level d is the merge, by
 * weight, of the leaves with the "packages" (sums of consecutive pairs) of level d + 1, a leaf going first on a
 * tie; the first 2n - 2 items of level 1 are taken, every package among the taken items of a level stands for two
 * items of the next one, and a leaf's code length is the number of levels in which it was taken. */
int orc_limit_lengths(const int64_t hist[ORC_NSYM], uint32_t length[ORC_NSYM], int limit) {
  int order[ORC_NSYM], n = 0, mx = 0;
  for (int i = 0; i < ORC_NSYM; i++)
    if (length[i]) {
      order[n++] = i;
      if ((int)length[i] > mx) mx = (int)length[i];
    }
  if (mx <= limit || n < 2) return mx;
  for (int a = 1; a < n; a++) { /* insertion sort: n <= 257 */
    int v = order[a], b = a - 1;
    while (b >= 0 && (hist[order[b]] > hist[v] || (hist[order[b]] == hist[v] && order[b] > v))) {
      order[b + 1] = order[b];
      b--;
    }
    order[b + 1] = v;
  }
  static uint64_t w[2][2 * ORC_NSYM];      /* weights of the current / previous level's list */
  static uint8_t is_leaf[66][2 * ORC_NSYM]; /* per level: item k is a leaf */
  static int len_of[66];
  int prev_n = 0, cur = 0;
  for (int d = limit; d >= 1; d--) {
    const uint64_t* pw = w[cur ^ 1];
    uint64_t* cw = w[cur];
    int npk = prev_n / 2, li = 0, pi = 0, k = 0;
    while (li < n || pi < npk) {
      uint64_t lw = li < n ? (uint64_t)hist[order[li]] : ~0ull;
      uint64_t pk = pi < npk ? pw[2 * pi] + pw[2 * pi + 1] : ~0ull;
      if (li < n && (pi >= npk || lw <= pk)) {
        cw[k] = lw;
        is_leaf[d][k] = 1;
        li++;
      } else {
        cw[k] = pk;
        is_leaf[d][k] = 0;
        pi++;
      }
      k++;
    }
    len_of[d] = k;
    prev_n = k;
    cur ^= 1;
  }
  int need = 2 * n - 2, taken[66];
  for (int d = 1; d <= limit; d++) {
    if (need > len_of[d]) need = len_of[d];
    int leaves = 0;
    for (int k = 0; k < need; k++) leaves += is_leaf[d][k];
    taken[d] = leaves;
    need = 2 * (need - leaves);
  }
  int newmax = 0;
  for (int i = 0; i < n; i++) {
    uint32_t l = 0;
    for (int d = 1; d <= limit; d++) l += i < taken[d];
    length[order[i]] = l;
    if ((int)l > newmax) newmax = (int)l;
  }
  return newmax;
}

/* orc_build_code with the opt-in limit: identical to orc_build_code whenever that one succeeds */
int orc_build_code_limited(const int64_t hist_in[ORC_NSYM], orc_code* c, int limit) {
  int64_t hist[ORC_NSYM];
  int nz = 0;
  memcpy(hist, hist_in, sizeof hist);
  for (int i = 0; i < 256; i++) nz += hist[i] != 0;
  if (nz == 0) return -1;
  memset(c, 0, sizeof *c);
  c->max_len = orc_code_lengths(hist, c->length); /* mutates hist: use the caller's counts for the ordering */
  if (c->max_len > limit) c->max_len = orc_limit_lengths(hist_in, c->length, limit);
  orc_canonical(c);
  return 0;
}

int orc_compress_limited(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_n, int limit) {
  int64_t hist[ORC_NSYM];
  orc_code c;
  orc_histogram(in, n, hist);
  int rc = orc_build_code_limited(hist, &c, limit);
  if (rc) return rc;
  size_t hs = orc_header_size(&c);
  if (cap < hs) return -3;
  orc_write_header(&c, out);
  size_t bs = orc_encode_body(in, n, &c, out + hs, cap - hs);
  if (bs == (size_t)-1) return -3;
  *out_n = hs + bs;
  return 0;
}
