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
