// golden-huffman_amd/host/ghf_tool.cc -- the flow of the reference's unit_tests/test.cc on the HIP policies,
// without gtest/Boost (neither is installed):
//   ghf_tool <file>           run the suite of unit_tests/test.cc:247-280 on <file>:
//                             canonical_huff_char.{compress_perf,decomress_perf,func},
//                             fast_canonical_huff_char.{decomress_perf,func}, table_canonical_huff_char.{...}
//                             (compress once, decompress with the three decoder names, byte-compare each time)
//                             and normal_huff_char.{compress_perf,decomress_perf,func} (:165-210, commented out there)
//   ghf_tool <file> 1         compress   <file> -> <file>.crs   (normal Huffman, unit_tests/test.cc:295-297)
//   ghf_tool <file> 2         decompress <file> -> <file>.de    (normal Huffman, :299-301)
//   ghf_tool <file> 3         compress   <file> -> <file>.crs2                 (unit_tests/test.cc:302-304)
//   ghf_tool <file> 4|5|6     decompress <file> -> <file>.de   (canonical | fast | table decoder, :306-314)
//   ghf_tool <file> 7         file-to-file throughput of the canonical pair: compress and decompress
//                             <file> three times each, byte-compare, print one JSON line (input bytes / wall time;
//                             the first run includes context creation, pinned rings and the I/O threads)
// Exit code 0 = everything matched.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "glzip_hip.h"

using namespace glzip_hip;

static std::string infile_name("5big.log");  // the reference's default, unit_tests/test.cc:38
static std::string outfile_name, infile_name2, outfile_name2;
static Compressor<HipNormalHuffEncoder<> > compressor;       // file-scope like unit_tests/test.cc:45
static Compressor<HipCanonicalHuffEncoder<> > compressor2;  // file-scope like unit_tests/test.cc:46

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// unit_tests/test.cc:48-84: every byte of the original equals the decompressed file, and the sizes match
static bool compressor_func_test() {
  FILE* a = fopen(infile_name.c_str(), "rb");
  FILE* b = fopen(outfile_name2.c_str(), "rb");
  if (!a || !b) {
    printf("  cannot reopen %s / %s\n", infile_name.c_str(), outfile_name2.c_str());
    return false;
  }
  static unsigned char ba[1 << 16], bb[1 << 16];
  unsigned long long pos = 0;
  bool ok = true;
  for (;;) {
    const size_t na = fread(ba, 1, sizeof ba, a), nb = fread(bb, 1, sizeof bb, b);
    if (na != nb) {
      printf("  file size differ around byte %llu\n", pos + (na < nb ? na : nb));
      ok = false;
      break;
    }
    if (na == 0) break;
    if (memcmp(ba, bb, na) != 0) {
      size_t i = 0;
      while (ba[i] == bb[i]) ++i;
      printf("  differ index is %llu: %u %u\n", pos + i, ba[i], bb[i]);
      ok = false;
      break;
    }
    pos += na;
  }
  fclose(a);
  fclose(b);
  return ok;
}

static void normal_huff_char_compress(const std::string& in) {  // unit_tests/test.cc:116-121
  outfile_name.clear();
  compressor.set_file(in, outfile_name);
  compressor.compress();
  compressor.clear();
}
static void canonical_huff_char_compress(const std::string& in) {  // unit_tests/test.cc:129-134
  outfile_name.clear();
  compressor2.set_file(in, outfile_name);
  compressor2.compress();
  compressor2.clear();
}
template <typename D>
static void decompress_with(const std::string& in) {  // unit_tests/test.cc:136-156
  outfile_name2.clear();
  Decompressor<D> d(in, outfile_name2);
  d.decompress();
}

template <typename F>
static bool run(const char* name, F f) {
  printf("[ RUN      ] %s\n", name);
  const double t0 = now_ms();
  bool ok = true;
  try {
    ok = f();
  } catch (const Error& e) {
    printf("  error %d: %s\n", e.status, e.what());
    ok = false;
  }
  printf("[ %s ] %s (%.1f ms)\n", ok ? "      OK" : " FAILED ", name, now_ms() - t0);
  return ok;
}

static int file_to_file_perf() {
  FILE* f = fopen(infile_name.c_str(), "rb");
  if (!f) {
    fprintf(stderr, "cannot open %s\n", infile_name.c_str());
    return 1;
  }
  fseek(f, 0, SEEK_END);
  const double gb = (double)ftell(f) / 1e9;
  fclose(f);
  double c_ms[3], d_ms[3];
  for (int i = 0; i < 3; ++i) {
    // fresh output files each time: truncating a multi-GiB file in /dev/shm costs as much as writing a third of it.
    // GHF_SINK=reuse: the files of the first round stay and are written over (their pages exist: the sink is then
    // bound by the copies, not by the kernel's page allocation)
    const char* how = getenv("GHF_SINK");
    if (!(how && strcmp(how, "reuse") == 0)) {
      remove((infile_name + ".crs2").c_str());
      remove((infile_name + ".crs2.de").c_str());
    }
    double t0 = now_ms();
    canonical_huff_char_compress(infile_name);
    c_ms[i] = now_ms() - t0;
    t0 = now_ms();
    decompress_with<HipCanonicalHuffDecoder<> >(outfile_name);
    d_ms[i] = now_ms() - t0;
  }
  const bool ok = compressor_func_test();
  const double c = c_ms[1] < c_ms[2] ? c_ms[1] : c_ms[2], d = d_ms[1] < d_ms[2] ? d_ms[1] : d_ms[2];
  printf("{\"file_GB\": %.6f, \"compress_ms\": [%.2f, %.2f, %.2f], \"decompress_ms\": [%.2f, %.2f, %.2f], "
         "\"compress_GBps\": %.3f, \"decompress_GBps\": %.3f, \"first_run_compress_GBps\": %.3f, \"round_trip_ok\": %s}\n",
         gb, c_ms[0], c_ms[1], c_ms[2], d_ms[0], d_ms[1], d_ms[2], gb / (c * 1e-3), gb / (d * 1e-3), gb / (c_ms[0] * 1e-3),
         ok ? "true" : "false");
  return ok ? 0 : 1;
}

int main(int argc, char* argv[]) {
  if (argc >= 2) infile_name = argv[1];
  try {
    if (argc == 3) {
      const int type = atoi(argv[2]);
      if (type == 7) return file_to_file_perf();
      if (type == 1) normal_huff_char_compress(infile_name);
      else if (type == 2) decompress_with<HipNormalHuffDecoder<> >(infile_name);
      else if (type == 3) canonical_huff_char_compress(infile_name);
      else if (type == 4) decompress_with<HipCanonicalHuffDecoder<> >(infile_name);
      else if (type == 5) decompress_with<HipFastCanonicalHuffDecoder<> >(infile_name);
      else if (type == 6) decompress_with<HipTableCanonicalHuffDecoder<> >(infile_name);
      else {
        fprintf(stderr, "mode must be 1..7\n");
        return 2;
      }
      return 0;
    }
  } catch (const Error& e) {
    fprintf(stderr, "ghf_tool: error %d: %s\n", e.status, e.what());
    return 1;
  }
  int failed = 0;
  failed += !run("canonical_huff_char.compress_perf", [] { canonical_huff_char_compress(infile_name); return true; });
  failed += !run("canonical_huff_char.decomress_perf", [] { decompress_with<HipCanonicalHuffDecoder<> >(outfile_name); return true; });
  failed += !run("canonical_huff_char.func", compressor_func_test);
  failed += !run("fast_canonical_huff_char.decomress_perf", [] { decompress_with<HipFastCanonicalHuffDecoder<> >(outfile_name); return true; });
  failed += !run("fast_canonical_huff_char.func", compressor_func_test);
  failed += !run("table_canonical_huff_char.decomress_perf", [] { decompress_with<HipTableCanonicalHuffDecoder<> >(outfile_name); return true; });
  failed += !run("table_canonical_huff_char.func", compressor_func_test);
  const std::string crs2 = outfile_name;
  failed += !run("normal_huff_char.compress_perf", [] { normal_huff_char_compress(infile_name); return true; });
  failed += !run("normal_huff_char.decomress_perf", [] { decompress_with<HipNormalHuffDecoder<> >(outfile_name); return true; });
  failed += !run("normal_huff_char.func", compressor_func_test);
  printf("[==========] 10 tests ran, %d failed. compressed files: %s %s\n", failed, crs2.c_str(), outfile_name.c_str());
  return failed ? 1 : 0;
}
