// oracle/ref_driver.cc -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A small driver around the *reference's own headers*, compiled where they lie under
// /root/reference (see oracle/Makefile, target `_ref`).  Nothing from the reference is
// copied into this repository: the headers are pulled in by -I at build time only and the
// resulting binary lands in oracle/_ref/ (git-ignored).
//
// It instantiates the byte-keyed canonical path exactly as unit_tests/test.cc:101-141 does
//   Compressor<CanonicalHuffEncoder<> >            (include/compressor.h:44-77)
//   Decompressor<CanonicalHuffDecoder<> >          (include/compressor.h:81-95)
// and additionally dumps the encoder's private tables (length_, codeword_, symbol_, ...,
// include/canonical_huff_encoder.h:107-120) so the golden fixtures can pin every stage.
//
// usage:  ref_glzip c  <in> <out.crs2>     compress
//         ref_glzip d  <in.crs2> <out>     decompress, bit-serial decoder (the only trustworthy one)
//         ref_glzip dt <in.crs2> <out>     decompress, TableCanonicalHuffDecoder  (run under timeout!)
//         ref_glzip df <in.crs2> <out>     decompress, FastCanonicalHuffDecoder   (run under timeout!)
//         ref_glzip t  <in>                dump stage tables as JSON on stdout (also writes <in>.crs2.tmp, removed)
//         ref_glzip b  <in> <out.crs2> <out.de>   time every phase, JSON on stdout
// SURVEY 8(f) N3, the .crs format (include/normal_huff_encoder.h, include/huff_tree.cc):
//         ref_glzip nc <in> <out.crs>      Compressor<NormalHuffEncoder<> >
//         ref_glzip nd <in.crs> <out>      Decompressor<NormalHuffDecoder<> >
//         ref_glzip nt <in>                dump the tree-order code strings as JSON on stdout
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <queue>
#include <deque>
#include <iostream>
#include <fstream>
#include <sstream>
#include <bitset>
#include <algorithm>
#include <functional>
#include <numeric>
#include <iomanip>
#include <typeinfo>
#include <chrono>
#include <memory.h>

// same trick as unit_tests/test.cc:21-24 (without DEBUG, which would pull gtest in)
#define private public
#define protected public
#include "compressor.h"
#include "canonical_huff_encoder.h"
#include "normal_huff_encoder.h"
#undef private
#undef protected

using namespace glzip;

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <typename T>
static void dump_arr(const char* name, const T* a, int n, bool last = false) {
  printf("\"%s\": [", name);
  for (int i = 0; i < n; i++) printf("%s%lld", i ? "," : "", (long long)a[i]);
  printf("]%s\n", last ? "" : ",");
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: see header\n"); return 2; }
  std::string mode = argv[1];
  std::string in = argv[2];
  std::string out = argc > 3 ? argv[3] : std::string();
  if (mode == "c") {
    Compressor<CanonicalHuffEncoder<> > c;
    c.set_file(in, out);
    c.compress();
    c.clear();
    return 0;
  }
  if (mode == "d") {
    Decompressor<CanonicalHuffDecoder<> > d(in, out);
    d.decompress();
    return 0;
  }
  if (mode == "dt") {
    Decompressor<TableCanonicalHuffDecoder<> > d(in, out);
    d.decompress();
    return 0;
  }
  if (mode == "df") {
    Decompressor<FastCanonicalHuffDecoder<> > d(in, out);
    d.decompress();
    return 0;
  }
  if (mode == "nc") {
    Compressor<NormalHuffEncoder<> > c;
    c.set_file(in, out);
    c.compress();
    c.clear();
    return 0;
  }
  if (mode == "nd") {
    Decompressor<NormalHuffDecoder<> > d(in, out);
    d.decompress();
    return 0;
  }
  if (mode == "nt") {
    std::string tmp = in + ".crs.tmp";
    NormalHuffEncoder<> e;
    e.set_file(in, tmp);
    e.caculate_frequency();
    e.gen_encode();
    printf("{\"codes\": [");
    for (int i = 0; i < 256; i++) printf("%s\"%s\"", i ? "," : "", e.encode_map_[i].c_str());
    printf("]}\n");
    e.clear();
    remove(tmp.c_str());
    return 0;
  }
  if (mode == "t") {
    std::string tmp = in + ".crs2.tmp";
    CanonicalHuffEncoder<> e;
    e.set_file(in, tmp);
    e.caculate_frequency();
    long long hist[CharSymbolNum];
    for (int i = 0; i < CharSymbolNum; i++) hist[i] = e.frequency_map_[i];
    e.gen_encode();
    printf("{\n");
    dump_arr("hist", hist, CharSymbolNum);
    dump_arr("length", e.length_, CharSymbolNum);
    // codeword_ is only assigned for symbols with length != 0 (canonical_huff_encoder.cc:127-133)
    unsigned int cw[CharSymbolNum];
    for (int i = 0; i < CharSymbolNum; i++) cw[i] = e.length_[i] ? e.codeword_[i] : 0;
    dump_arr("codeword", cw, CharSymbolNum);
    dump_arr("symbol", e.symbol_, CharSymbolNum);
    dump_arr("first_code", e.first_code_ + 1, e.max_len_);
    dump_arr("start_pos", e.start_pos_ + 1, e.max_len_);
    printf("\"min_len\": %d,\n\"max_len\": %d\n}\n", e.min_len_, e.max_len_);
    e.clear();
    remove(tmp.c_str());
    return 0;
  }
  if (mode == "b") {
    if (argc < 5) return 2;
    std::string de = argv[4];
    CanonicalHuffEncoder<> e;
    e.set_file(in, out);
    double t0 = now_s();
    e.caculate_frequency();
    double t1 = now_s();
    e.gen_encode();
    double t2 = now_s();
    e.write_encode_info();
    double t3 = now_s();
    e.encode_file();
    double t4 = now_s();
    e.clear();
    double t5, t6;
    {
      Decompressor<CanonicalHuffDecoder<> > d(out, de);
      t5 = now_s();
      d.decompress();
      t6 = now_s();
    }
    printf("{\"histogram_s\": %.6f, \"gen_encode_s\": %.6f, \"header_s\": %.6f, \"encode_file_s\": %.6f, "
           "\"encode_total_s\": %.6f, \"decode_bitserial_s\": %.6f}\n",
           t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0, t6 - t5);
    return 0;
  }
  fprintf(stderr, "unknown mode %s\n", mode.c_str());
  return 2;
}
