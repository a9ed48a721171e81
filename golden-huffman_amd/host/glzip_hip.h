// golden-huffman_amd/host/glzip_hip.h -- C++ host layer: the reference's Compressor<Encoder> /
// Decompressor<Decoder> API with policy classes that run on the MI355X through the C ABI (include/ghf.h).
//
// Mirrors, member for member (same names -- including the reference's spelling caculate_frequency --, same
// argument meaning, same file-name side effects):
//   Compressor<_Encoder>      include/compressor.h:44-77   (ctor(in, out&), default ctor, set_file, clear, compress)
//   Decompressor<_Decoder>    include/compressor.h:81-95   (ctor(in, out&), decompress)
//   CanonicalHuffEncoder<>    include/canonical_huff_encoder.h:48-121, include/encoder.h:56-213
//   CanonicalHuffDecoder<> / FastCanonicalHuffDecoder<> / TableCanonicalHuffDecoder<>
//                             include/canonical_huff_encoder.h:126-209, include/encoder.h:218-241
// so that   Compressor<HipCanonicalHuffEncoder<> > c; c.set_file(in, out); c.compress();
// writes the same bytes to the same file name as the reference's Compressor<CanonicalHuffEncoder<> >.
//
// Differences, all on the side of doing MORE than the reference:
//   * failures (unopenable file, empty input, code > 32 bits, corrupt stream, HIP error) throw
//     glzip_hip::Error; the reference has no error path at all (include/encoder.h:67-70 FIXME);
//   * file I/O goes through one pinned host buffer and hipMemcpyAsync instead of 64 KiB stdio buffers
//     (utils/include/buffer.h:61-317);
//   * only the byte-keyed (unsigned char) instantiation exists; anything else is a compile error.
// No HIP header is needed to compile this file: it links against libghf.so only.
#ifndef GLZIP_HIP_H_
#define GLZIP_HIP_H_
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdexcept>
#include <string>
#include <vector>

#include "ghf.h"

namespace glzip_hip {

struct Error : public std::runtime_error {
  int status;
  Error(int s, const std::string& what) : std::runtime_error(what), status(s) {}
};

namespace detail {

// one ghf_ctx per policy object ("one context per host thread/GPU"), created on first use so that a
// default-constructed Compressor (file-scope objects in unit_tests/test.cc:45-46) touches no GPU
class Session {
 public:
  Session() : ctx_(NULL) {}
  ~Session() {
    if (ctx_) ghf_ctx_destroy(ctx_);
  }
  ghf_ctx* ctx() const {
    if (!ctx_) {
      int rc = ghf_ctx_create(device_from_env(), &ctx_);
      if (rc) throw Error(rc, std::string("ghf_ctx_create: ") + ghf_status_string(rc) + " " + ghf_last_error(NULL));
    }
    return ctx_;
  }
  void check(int rc, const char* where) const {
    if (rc) throw Error(rc, std::string(where) + ": " + ghf_status_string(rc) + " " + ghf_last_error(ctx_));
  }
  void sync(const char* where) const {
    int rc = ghf_sync(ctx());
    if (rc) {
      ghf_clear_status(ctx_);
      throw Error(rc, std::string(where) + ": " + ghf_status_string(rc));
    }
  }
  static int device_from_env() {
    const char* e = getenv("GHF_DEVICE");
    return e ? atoi(e) : 0;
  }

 private:
  Session(const Session&);
  Session& operator=(const Session&);
  mutable ghf_ctx* ctx_;
};

struct DeviceBuf {
  const Session* s;
  void* p;
  size_t n;
  DeviceBuf() : s(NULL), p(NULL), n(0) {}
  ~DeviceBuf() { reset(); }
  void reset() {
    if (p) ghf_device_free(s->ctx(), p);
    p = NULL;
    n = 0;
  }
  void alloc(const Session& ss, size_t bytes) {
    reset();
    s = &ss;
    ss.check(ghf_device_alloc(ss.ctx(), bytes, &p), "ghf_device_alloc");
    n = bytes;
  }
  uint8_t* u8() const { return static_cast<uint8_t*>(p); }
};

struct PinnedBuf {
  const Session* s;
  void* p;
  size_t n;
  PinnedBuf() : s(NULL), p(NULL), n(0) {}
  ~PinnedBuf() { reset(); }
  void reset() {
    if (p) ghf_host_free(s->ctx(), p);
    p = NULL;
    n = 0;
  }
  void alloc(const Session& ss, size_t bytes) {
    reset();
    s = &ss;
    ss.check(ghf_host_alloc(ss.ctx(), bytes, &p), "ghf_host_alloc");
    n = bytes;
  }
  uint8_t* u8() const { return static_cast<uint8_t*>(p); }
};

inline size_t file_size(FILE* f) {
  fseek(f, 0, SEEK_END);
  long long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  return n < 0 ? 0 : (size_t)n;
}

// What replaces the reference's 64 KiB FixedFileBuffer (utils/include/buffer.h:61-317): files move between disk
// and HBM in 32 MiB pieces through TWO pinned buffers, each with its own copy stream, so the fread/fwrite of one
// piece overlaps the hipMemcpyAsync of the other.
class Stager {
 public:
  static const size_t kPiece = 32u << 20;
  Stager() {}
  // file -> device, n bytes from the current file position
  void to_device(FILE* f, uint8_t* d_dst, size_t n, const std::string& what) {
    ensure();
    size_t off = 0;
    for (int k = 0; off < n; ++k) {
      const int b = k & 1;
      const size_t len = n - off < kPiece ? n - off : kPiece;
      s_[b].sync("stager");  // the copy that last used this buffer is done
      if (fread(h_[b].p, 1, len, f) != len) throw Error(GHF_E_INVAL, "short read on " + what);
      s_[b].check(ghf_copy_h2d(s_[b].ctx(), d_dst + off, h_[b].p, len), "ghf_copy_h2d");
      off += len;
    }
    s_[0].sync("stager");
    s_[1].sync("stager");
  }
  // device -> file, n bytes appended at the current file position
  void to_file(const uint8_t* d_src, size_t n, FILE* f, const std::string& what) {
    ensure();
    size_t off = 0, pending_len[2] = {0, 0};
    for (int k = 0; off < n || pending_len[k & 1] || pending_len[(k + 1) & 1]; ++k) {
      const int b = k & 1;
      if (pending_len[b]) {  // the copy issued two rounds ago into this buffer: wait, then write it out
        s_[b].sync("stager");
        if (fwrite(h_[b].p, 1, pending_len[b], f) != pending_len[b]) throw Error(GHF_E_INVAL, "short write on " + what);
        pending_len[b] = 0;
      }
      if (off < n) {
        const size_t len = n - off < kPiece ? n - off : kPiece;
        s_[b].check(ghf_copy_d2h(s_[b].ctx(), h_[b].p, d_src + off, len), "ghf_copy_d2h");
        pending_len[b] = len;
        off += len;
      }
    }
  }

 private:
  void ensure() {
    for (int b = 0; b < 2; ++b)
      if (!h_[b].p) h_[b].alloc(s_[b], kPiece);
  }
  Session s_[2];
  PinnedBuf h_[2];
};

}  // namespace detail

// ------------------------------------------------------------------------------------------------ Compressor
template <typename _Encoder>
class Compressor {  // include/compressor.h:44-77
 public:
  Compressor(const std::string& infile_name, std::string& outfile_name) : encoder_(infile_name, outfile_name) {}
  Compressor() {}
  void set_file(const std::string& infile_name, std::string& outfile_name) { encoder_.set_file(infile_name, outfile_name); }
  void clear() { encoder_.clear(); }
  void compress() {  // include/compressor.h:62-73: the four steps, in this order
    encoder_.caculate_frequency();
    encoder_.gen_encode();
    encoder_.write_encode_info();
    encoder_.encode_file();
  }
  _Encoder& encoder() { return encoder_; }

 private:
  _Encoder encoder_;
};

template <typename _Decoder>
class Decompressor {  // include/compressor.h:81-95
 public:
  Decompressor(const std::string& infile_name, std::string& outfile_name) : decoder_(infile_name, outfile_name) {}
  void decompress() {
    decoder_.get_encode_info();
    decoder_.decode_file();
  }
  _Decoder& decoder() { return decoder_; }

 private:
  _Decoder decoder_;
};

// ------------------------------------------------------------------------------------------------ encoder policy
template <typename _KeyType = unsigned char>
class HipCanonicalHuffEncoder;

template <>
class HipCanonicalHuffEncoder<unsigned char> {
 public:
  HipCanonicalHuffEncoder(const std::string& infile_name, std::string& outfile_name) : infile_(NULL), outfile_(NULL), n_(0) {
    set_file(infile_name, outfile_name);
  }
  HipCanonicalHuffEncoder() : infile_(NULL), outfile_(NULL), n_(0) {}
  ~HipCanonicalHuffEncoder() { clear(); }

  // include/canonical_huff_encoder.cc:15-32: the output name defaults to <in>.crs2 and is handed back
  void set_file(const std::string& infile_name, std::string& outfile_name) {
    clear();
    infile_name_ = infile_name;
    infile_ = fopen(infile_name.c_str(), "rb");
    if (!infile_) throw Error(GHF_E_INVAL, "cannot open input file " + infile_name);
    if (outfile_name.empty()) outfile_name = infile_name + ".crs2";
    outfile_ = fopen(outfile_name.c_str(), "wb");
    if (!outfile_) throw Error(GHF_E_INVAL, "cannot open output file " + outfile_name);
  }

  void clear() {  // include/encoder.h:85-92
    if (infile_) fclose(infile_);
    if (outfile_) fclose(outfile_);
    infile_ = NULL;
    outfile_ = NULL;
  }

  // include/encoder.h:99-105,123-150: the file goes to the GPU once and stays there for encode_file
  void caculate_frequency() {
    n_ = detail::file_size(infile_);
    if (n_ == 0) throw Error(GHF_E_EMPTY, "empty input: undefined in the reference, refused here");
    d_in_.alloc(s_, n_ + 16);
    d_hist_.alloc(s_, GHF_NSYM * sizeof(uint64_t));
    stager_.to_device(infile_, d_in_.u8(), n_, infile_name_);  // pieces: fread overlaps hipMemcpyAsync
    s_.check(ghf_histogram(s_.ctx(), d_in_.u8(), n_, static_cast<uint64_t*>(d_hist_.p)), "ghf_histogram");
  }

  // include/canonical_huff_encoder.cc:35-42
  void gen_encode() {
    d_code_.alloc(s_, sizeof(ghf_code));
    // opt-in: set_code_limit(true) or GHF_CODE_LIMIT=1 in the environment replaces the reference's
    // "undefined above 32 bits" by the optimal 32-bit-limited code; without it the behaviour is the reference's
    const char* env = getenv("GHF_CODE_LIMIT");
    const unsigned flags = (limit_ || (env && env[0] == '1')) ? GHF_CODE_LIMIT : 0u;
    s_.check(ghf_build_code_ex(s_.ctx(), static_cast<const uint64_t*>(d_hist_.p), static_cast<ghf_code*>(d_code_.p), flags),
             "ghf_build_code");
    s_.check(ghf_copy_d2h(s_.ctx(), &code_, d_code_.p, sizeof(ghf_code)), "ghf_copy_d2h");
    s_.sync("gen_encode");
  }

  // include/canonical_huff_encoder.cc:210-242: header at file offset 0, flushed before the body is produced
  void write_encode_info() {
    const size_t hdr = ghf_header_bytes(code_.max_len);
    cap_ = ghf_compress_bound(n_);
    d_out_.alloc(s_, cap_);
    h_out_.alloc(s_, 2048);  // the header only (<= 1296 bytes); the body is streamed through the stager
    s_.check(ghf_write_header(s_.ctx(), static_cast<const ghf_code*>(d_code_.p), d_out_.u8(), cap_), "ghf_write_header");
    s_.check(ghf_copy_d2h(s_.ctx(), h_out_.p, d_out_.p, hdr), "ghf_copy_d2h");
    s_.sync("write_encode_info");
    fseek(outfile_, 0, SEEK_SET);
    if (fwrite(h_out_.p, 1, hdr, outfile_) != hdr) throw Error(GHF_E_INVAL, "short write (header)");
    fflush(outfile_);
  }

  // include/canonical_huff_encoder.cc:245-285
  void encode_file() {
    const size_t hdr = ghf_header_bytes(code_.max_len);
    detail::DeviceBuf d_end;
    d_end.alloc(s_, 2 * sizeof(uint64_t));
    uint64_t end[2] = {0, 0};
    const ghf_code* dc = static_cast<const ghf_code*>(d_code_.p);
    s_.check(ghf_encode_plan(s_.ctx(), d_in_.u8(), n_, dc, NULL), "ghf_encode_plan");
    s_.check(ghf_encode_emit(s_.ctx(), d_in_.u8(), n_, dc, NULL, GHF_EMIT_LAST, d_out_.u8(), cap_, NULL,
                             static_cast<uint64_t*>(d_end.p)),
             "ghf_encode_emit");
    s_.check(ghf_copy_d2h(s_.ctx(), end, d_end.p, sizeof end), "ghf_copy_d2h");
    s_.sync("encode_file");
    const size_t total = (size_t)end[1];
    stager_.to_file(d_out_.u8() + hdr, total - hdr, outfile_, "output (body)");  // pieces: fwrite overlaps the D2H copies
    fflush(outfile_);
  }

  const ghf_code& code() const { return code_; }  // length_/codeword_/symbol_/... of the reference, for tests
  void set_code_limit(bool on) { limit_ = on; }   // not in the reference: see gen_encode()

 private:
  HipCanonicalHuffEncoder(const HipCanonicalHuffEncoder&);
  HipCanonicalHuffEncoder& operator=(const HipCanonicalHuffEncoder&);
  detail::Session s_;
  FILE* infile_;
  FILE* outfile_;
  std::string infile_name_;
  size_t n_, cap_;
  bool limit_ = false;
  detail::PinnedBuf h_out_;
  detail::DeviceBuf d_in_, d_hist_, d_code_, d_out_;
  detail::Stager stager_;
  ghf_code code_;
};

// ------------------------------------------------------------------------------------------------ decoder policy
template <typename _KeyType = unsigned char>
class HipCanonicalHuffDecoder;

template <>
class HipCanonicalHuffDecoder<unsigned char> {
 public:
  // include/encoder.h:227-232: the output name defaults to <in>.de and is handed back
  HipCanonicalHuffDecoder(const std::string& infile_name, std::string& outfile_name) : hdr_(0) {
    infile_ = fopen(infile_name.c_str(), "rb");
    if (!infile_) throw Error(GHF_E_INVAL, "cannot open input file " + infile_name);
    if (outfile_name.empty()) outfile_name = infile_name + ".de";
    outfile_ = fopen(outfile_name.c_str(), "wb");
    if (!outfile_) {
      fclose(infile_);
      throw Error(GHF_E_INVAL, "cannot open output file " + outfile_name);
    }
  }
  ~HipCanonicalHuffDecoder() {
    if (infile_) fclose(infile_);
    if (outfile_) fclose(outfile_);
  }

  // include/canonical_huff_encoder.cc:349-374 -- plus the validation the reference does not do
  void get_encode_info() {
    n_ = detail::file_size(infile_);
    std::vector<uint8_t> head(n_ < 1296 ? n_ : 1296);  // the largest header: 1040 + 8 * 32
    if (fread(head.data(), 1, head.size(), infile_) != head.size()) throw Error(GHF_E_INVAL, "short read");
    s_.check(ghf_parse_header(head.data(), head.size(), &code_, &hdr_), "ghf_parse_header");
    fseek(infile_, 0, SEEK_SET);
  }

  // include/canonical_huff_encoder.cc:377-419 (and :422-461, :519-568): runs until the end mark
  void decode_file() {
    d_in_.alloc(s_, n_ + 16);
    d_code_.alloc(s_, sizeof(ghf_code));
    stager_.to_device(infile_, d_in_.u8(), n_, "input");
    s_.check(ghf_copy_h2d(s_.ctx(), d_code_.p, &code_, sizeof(ghf_code)), "ghf_copy_h2d");
    const ghf_code* dc = static_cast<const ghf_code*>(d_code_.p);
    uint64_t n_out = 0;
    s_.check(ghf_decoded_size(s_.ctx(), d_in_.u8(), n_, dc, &n_out), "ghf_decoded_size");
    d_out_.alloc(s_, (size_t)n_out + 16);
    s_.check(ghf_decode(s_.ctx(), d_in_.u8(), n_, dc, NULL, d_out_.u8(), (size_t)n_out + 16, NULL), "ghf_decode");
    s_.sync("decode_file");
    stager_.to_file(d_out_.u8(), (size_t)n_out, outfile_, "output");
    fflush(outfile_);
  }

 private:
  HipCanonicalHuffDecoder(const HipCanonicalHuffDecoder&);
  HipCanonicalHuffDecoder& operator=(const HipCanonicalHuffDecoder&);
  detail::Session s_;
  FILE* infile_;
  FILE* outfile_;
  size_t n_, hdr_;
  detail::DeviceBuf d_in_, d_code_, d_out_;
  detail::Stager stager_;
  ghf_code code_;
};

// The reference has three interchangeable decoders that all emit the same bytes (bit-serial, left-justified
// linear search, 8-bit length table).  On the GPU they are one kernel (direct table + the linear extension as
// its fallback), so the other two names are the same class.
template <typename _KeyType = unsigned char>
class HipFastCanonicalHuffDecoder : public HipCanonicalHuffDecoder<_KeyType> {
 public:
  HipFastCanonicalHuffDecoder(const std::string& in, std::string& out) : HipCanonicalHuffDecoder<_KeyType>(in, out) {}
};
template <typename _KeyType = unsigned char, int TableLength = 8>
class HipTableCanonicalHuffDecoder : public HipCanonicalHuffDecoder<_KeyType> {
 public:
  HipTableCanonicalHuffDecoder(const std::string& in, std::string& out) : HipCanonicalHuffDecoder<_KeyType>(in, out) {}
};

// ================================================================================================
// the .crs format.  Same member functions, same order (Compressor<>::compress() and
// Decompressor<>::decompress() above do not change): drop-ins for NormalHuffEncoder<> / NormalHuffDecoder<>
// (include/normal_huff_encoder.h:57-283).
// ================================================================================================
template <typename _KeyType = unsigned char>
class HipNormalHuffEncoder;

template <>
class HipNormalHuffEncoder<unsigned char> {
 public:
  HipNormalHuffEncoder(const std::string& infile_name, std::string& outfile_name) : infile_(NULL), outfile_(NULL), n_(0) {
    set_file(infile_name, outfile_name);
  }
  HipNormalHuffEncoder() : infile_(NULL), outfile_(NULL), n_(0) {}
  ~HipNormalHuffEncoder() { clear(); }

  // include/normal_huff_encoder.h:83-99: the output name defaults to <in>.crs and is handed back
  void set_file(const std::string& infile_name, std::string& outfile_name) {
    clear();
    infile_name_ = infile_name;
    infile_ = fopen(infile_name.c_str(), "rb");
    if (!infile_) throw Error(GHF_E_INVAL, "cannot open input file " + infile_name);
    if (outfile_name.empty()) outfile_name = infile_name + ".crs";
    outfile_ = fopen(outfile_name.c_str(), "wb");
    if (!outfile_) throw Error(GHF_E_INVAL, "cannot open output file " + outfile_name);
  }

  void clear() {  // include/encoder.h:85-92
    if (infile_) fclose(infile_);
    if (outfile_) fclose(outfile_);
    infile_ = NULL;
    outfile_ = NULL;
  }

  // include/encoder.h:99-105,136-150 (no end-mark slot in this format: init_nhuff, normal_huff_encoder.h:189-196)
  void caculate_frequency() {
    n_ = detail::file_size(infile_);
    if (n_ == 0) throw Error(GHF_E_EMPTY, "empty input: undefined in the reference, refused here");
    d_in_.alloc(s_, n_ + 16);
    d_hist_.alloc(s_, GHF_NSYM * sizeof(uint64_t));
    stager_.to_device(infile_, d_in_.u8(), n_, infile_name_);
    s_.check(ghf_histogram(s_.ctx(), d_in_.u8(), n_, static_cast<uint64_t*>(d_hist_.p)), "ghf_histogram");
  }

  // include/normal_huff_encoder.h:110-121 -> EncodeHuffTree::build_tree + gen_encode (include/huff_tree.cc:138-171)
  void gen_encode() {
    d_tree_.alloc(s_, sizeof(ghf_tree));
    d_code_.alloc(s_, sizeof(ghf_code));
    s_.check(ghf_crs_build_code(s_.ctx(), static_cast<const uint64_t*>(d_hist_.p), static_cast<ghf_tree*>(d_tree_.p),
                                static_cast<ghf_code*>(d_code_.p)),
             "ghf_crs_build_code");
    s_.check(ghf_copy_d2h(s_.ctx(), &tree_, d_tree_.p, sizeof(ghf_tree)), "ghf_copy_d2h");
    s_.sync("gen_encode");
  }

  // include/normal_huff_encoder.h:136-138 -> serialize_tree (include/huff_tree.cc:174-187): the tree at file offset 0
  void write_encode_info() {
    fseek(outfile_, 0, SEEK_SET);
    if (fwrite(tree_.header, 1, tree_.tree_bytes, outfile_) != tree_.tree_bytes) throw Error(GHF_E_INVAL, "short write (tree)");
    fflush(outfile_);
  }

  // include/normal_huff_encoder.h:159-186: {left_bits, last byte}, then the whole bytes of the body
  void encode_file() {
    const size_t hdr = (size_t)tree_.tree_bytes + 2;
    cap_ = ghf_crs_compress_bound(n_);
    d_out_.alloc(s_, cap_);
    detail::DeviceBuf d_scal;
    d_scal.alloc(s_, 4 * sizeof(uint64_t));  // [0] start bit, [1] total bits, [2..3] end
    uint64_t* ds = static_cast<uint64_t*>(d_scal.p);
    const uint64_t start_bit = 8ull * hdr;
    uint64_t scal[4] = {start_bit, 0, 0, 0};
    const ghf_code* dc = static_cast<const ghf_code*>(d_code_.p);
    s_.check(ghf_copy_h2d(s_.ctx(), ds, scal, sizeof(uint64_t)), "ghf_copy_h2d");
    s_.check(ghf_encode_plan(s_.ctx(), d_in_.u8(), n_, dc, ds + 1), "ghf_encode_plan");
    s_.check(ghf_encode_emit(s_.ctx(), d_in_.u8(), n_, dc, ds, 0, d_out_.u8(), cap_, NULL, ds + 2), "ghf_encode_emit");
    s_.check(ghf_copy_d2h(s_.ctx(), scal, ds, sizeof scal), "ghf_copy_d2h");
    s_.sync("encode_file");
    const uint64_t bits = scal[1];
    const size_t whole = (size_t)(bits >> 3);
    const unsigned left = (unsigned)((8 - (bits & 7)) & 7);
    unsigned char prefix[2] = {(unsigned char)left, 0};
    if (left) {  // the zero-filled last byte sits right behind the whole bytes (the emit kernels zero-fill their last unit)
      s_.check(ghf_copy_d2h(s_.ctx(), prefix + 1, d_out_.u8() + hdr + whole, 1), "ghf_copy_d2h");
      s_.sync("encode_file (last byte)");
    }
    if (fwrite(prefix, 1, 2, outfile_) != 2) throw Error(GHF_E_INVAL, "short write (prefix)");
    stager_.to_file(d_out_.u8() + hdr, whole, outfile_, "output (body)");
    fflush(outfile_);
  }

  const ghf_tree& tree() const { return tree_; }

 private:
  HipNormalHuffEncoder(const HipNormalHuffEncoder&);
  HipNormalHuffEncoder& operator=(const HipNormalHuffEncoder&);
  detail::Session s_;
  FILE* infile_;
  FILE* outfile_;
  std::string infile_name_;
  size_t n_, cap_;
  detail::DeviceBuf d_in_, d_hist_, d_tree_, d_code_, d_out_;
  detail::Stager stager_;
  ghf_tree tree_;
};

template <typename _KeyType = unsigned char>
class HipNormalHuffDecoder;

template <>
class HipNormalHuffDecoder<unsigned char> {
 public:
  // include/encoder.h:227-232: the output name defaults to <in>.de and is handed back
  HipNormalHuffDecoder(const std::string& infile_name, std::string& outfile_name) : n_(0), tb_(0), left_(0), last_(0) {
    infile_ = fopen(infile_name.c_str(), "rb");
    if (!infile_) throw Error(GHF_E_INVAL, "cannot open input file " + infile_name);
    if (outfile_name.empty()) outfile_name = infile_name + ".de";
    outfile_ = fopen(outfile_name.c_str(), "wb");
    if (!outfile_) {
      fclose(infile_);
      throw Error(GHF_E_INVAL, "cannot open output file " + outfile_name);
    }
  }
  ~HipNormalHuffDecoder() {
    if (infile_) fclose(infile_);
    if (outfile_) fclose(outfile_);
  }

  // include/normal_huff_encoder.h:268-271 -> DecodeHuffTree::build_tree (include/huff_tree.cc:289-303), validated
  void get_encode_info() {
    n_ = detail::file_size(infile_);
    std::vector<uint8_t> head(n_ < 1024 ? n_ : 1024);  // the largest tree: 2 * 511 bytes, then the two prefix bytes
    if (fread(head.data(), 1, head.size(), infile_) != head.size()) throw Error(GHF_E_INVAL, "short read");
    s_.check(ghf_crs_parse_header(head.data(), head.size(), &tree_, &tb_), "ghf_crs_parse_header");
    if (tb_ + 2 > head.size()) throw Error(GHF_E_FORMAT, "the two bytes behind the tree are missing");
    left_ = head[tb_];       // include/huff_tree.cc:195-196
    last_ = head[tb_ + 1];
    if (left_ > 7) throw Error(GHF_E_FORMAT, "left_bits > 7");
    fseek(infile_, 0, SEEK_SET);
  }

  // include/normal_huff_encoder.h:272-274 -> DecodeHuffTree::decode_file (include/huff_tree.cc:191-207)
  void decode_file() {
    d_in_.alloc(s_, n_ + 32);
    d_tree_.alloc(s_, sizeof(ghf_tree));
    stager_.to_device(infile_, d_in_.u8(), n_, "input");
    size_t stream_bytes = n_;
    if (left_) {  // the stored last byte goes behind the body, where its bits belong
      s_.check(ghf_copy_h2d(s_.ctx(), d_in_.u8() + n_, &last_, 1), "ghf_copy_h2d");
      ++stream_bytes;
    }
    s_.check(ghf_copy_h2d(s_.ctx(), d_tree_.p, &tree_, sizeof(ghf_tree)), "ghf_copy_h2d");
    const ghf_tree* dt = static_cast<const ghf_tree*>(d_tree_.p);
    uint64_t n_out = 0;
    s_.check(ghf_crs_decoded_size(s_.ctx(), d_in_.u8(), stream_bytes, (int)left_, dt, &n_out), "ghf_crs_decoded_size");
    d_out_.alloc(s_, (size_t)n_out + 16);
    s_.check(ghf_crs_decode(s_.ctx(), d_in_.u8(), stream_bytes, (int)left_, dt, NULL, d_out_.u8(), (size_t)n_out + 16, NULL),
             "ghf_crs_decode");
    s_.sync("decode_file");
    stager_.to_file(d_out_.u8(), (size_t)n_out, outfile_, "output");
    fflush(outfile_);
  }

 private:
  HipNormalHuffDecoder(const HipNormalHuffDecoder&);
  HipNormalHuffDecoder& operator=(const HipNormalHuffDecoder&);
  detail::Session s_;
  FILE* infile_;
  FILE* outfile_;
  size_t n_, tb_;
  uint8_t left_, last_;
  detail::DeviceBuf d_in_, d_tree_, d_out_;
  detail::Stager stager_;
  ghf_tree tree_;
};

}  // namespace glzip_hip
#endif  // GLZIP_HIP_H_
