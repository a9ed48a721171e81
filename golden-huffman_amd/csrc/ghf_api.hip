// golden-huffman_amd/csrc/ghf_api.hip -- the C ABI of include/ghf.h on top of the gfx950 kernels.
// Host-side only: argument checks, workspace, launches.  No compute happens on the CPU here except
// ghf_parse_header (a <= 1.3 KiB header; SURVEY 8 row a7 keeps it on the host).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "ghf_internal.h"

using namespace ghf;

struct ghf_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  int* d_status = nullptr;
  int* h_status = nullptr;  // pinned
  // workspace
  uint32_t* d_chunk_hist = nullptr;
  size_t chunk_hist_cap = 0;  // in chunks
  uint64_t* d_chunk_off = nullptr;
  size_t chunk_off_cap = 0;  // in entries
  const uint8_t* hist_in = nullptr;  // what d_chunk_hist currently describes
  uint64_t hist_n = 0;
  uint32_t hist_chunk = 0;
  const uint8_t* plan_in = nullptr;  // what d_chunk_off currently describes
  uint64_t plan_n = 0;
  const ghf_code* plan_code = nullptr;
  uint64_t* d_hist = nullptr;   // [257]
  uint64_t* d_hist_acc = nullptr;  // K1's replicated totals + arrival counter, zero between launches
  ghf_code* d_code = nullptr;   // scratch tables for ghf_compress
  ghf_tree* d_tree = nullptr;   // scratch tree for ghf_crs_compress
  DecTables* d_dt = nullptr;
  const ghf_code* dt_code = nullptr;  // ghf_decode_prepare() built d_dt from these tables; consumed by the next ghf_decode
  uint64_t* d_totals = nullptr;  // [totals_cap] per-rank body bits (ghf_encode_sharded)
  int totals_cap = 0;
  uint64_t* d_u64 = nullptr;    // [16] scratch scalars: 0 total_bits, 1..2 end, 3 n_symbols, 4 eof_sub, 6 start bit (.crs), 8 landing, 9 changed + count, 10 first moved (inverted)
  uint64_t* h_u64 = nullptr;    // [16] pinned mirror
  // K6 workspace (foreign streams)
  void* d_sync = nullptr;
  size_t sync_cap = 0;  // bytes
  ghf_index fidx = {};  // side-car rebuilt for the last foreign stream
  uint64_t* d_seg_abs = nullptr;
  size_t fidx_cap_segs = 0, fidx_cap_chunks = 0;
  const uint8_t* fidx_stream = nullptr;  // which stream c->fidx currently describes
  size_t fidx_bytes = 0;
  std::string err;
};

namespace {

thread_local std::string g_create_err;  // ghf_last_error(NULL): why ghf_ctx_create failed

int fail(ghf_ctx* c, int code, const char* what, hipError_t e = hipSuccess) {
  if (c) {
    c->err = what;
    if (e != hipSuccess) {
      c->err += ": ";
      c->err += hipGetErrorString(e);
    }
  }
  return code;
}

#define GHF_HIP(c, call)                                         \
  do {                                                           \
    hipError_t e_ = (call);                                      \
    if (e_ != hipSuccess) return fail((c), GHF_E_HIP, #call, e_); \
  } while (0)

int ensure_ws(ghf_ctx* c, size_t nchunks) {
  if (nchunks + 1 > c->chunk_off_cap) {
    if (c->d_chunk_off) (void)hipFree(c->d_chunk_off);
    c->d_chunk_off = nullptr;
    c->chunk_off_cap = 0;
    size_t cap = std::max<size_t>(nchunks + 1, 1024);
    GHF_HIP(c, hipMalloc(&c->d_chunk_off, cap * sizeof(uint64_t)));
    c->chunk_off_cap = cap;
    c->plan_in = nullptr;
  }
  if (nchunks > c->chunk_hist_cap) {
    if (c->d_chunk_hist) (void)hipFree(c->d_chunk_hist);
    c->d_chunk_hist = nullptr;
    c->chunk_hist_cap = 0;
    size_t cap = std::max<size_t>(nchunks, 1024);
    GHF_HIP(c, hipMalloc(&c->d_chunk_hist, cap * 256 * sizeof(uint32_t)));
    c->chunk_hist_cap = cap;
    c->hist_in = nullptr;
  }
  return GHF_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

// accessors for ghf_comm.hip (the struct stays private to this file)
int ghf_api_fail(ghf_ctx* c, int code, const char* what) { return fail(c, code, what); }
hipStream_t ghf_api_stream(ghf_ctx* c) { return c->stream; }
int ghf_api_device(ghf_ctx* c) { return c->device; }
uint64_t* ghf_api_scratch_u64(ghf_ctx* c) { return c->d_u64; }
uint64_t* ghf_api_hist(ghf_ctx* c) { return c->d_hist; }
uint64_t* ghf_api_totals(ghf_ctx* c, int world) {
  if (world > c->totals_cap) {
    if (c->d_totals) (void)hipFree(c->d_totals);
    c->d_totals = nullptr;
    c->totals_cap = 0;
    if (hipMalloc(&c->d_totals, (size_t)(world < 8 ? 8 : world) * sizeof(uint64_t)) != hipSuccess) return nullptr;
    c->totals_cap = world < 8 ? 8 : world;
  }
  return c->d_totals;
}

extern "C" {

int ghf_version(void) { return 210; }  // 2.1: events, ghf_histogram_add, GHF_EMPTY_OK, table validation

const char* ghf_status_string(int s) {
  switch (s) {
    case GHF_OK: return "ok";
    case GHF_E_INVAL: return "invalid argument";
    case GHF_E_HIP: return "HIP runtime error / no device";
    case GHF_E_EMPTY: return "empty input (undefined in the reference)";
    case GHF_E_CODELEN: return "code longer than 32 bits (reference limit)";
    case GHF_E_CAP: return "output capacity too small";
    case GHF_E_FORMAT: return "not a .crs2 / .crs header";
    case GHF_E_CORRUPT: return "corrupt stream";
    case GHF_E_NOMEM: return "out of memory";
    case GHF_E_SINGLE: return "one distinct byte value (.crs: undefined in the reference)";
    default: return "unknown status";
  }
}

int ghf_ctx_create(int device, ghf_ctx** out) {
  if (!out) return GHF_E_INVAL;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {  // no CPU fallback, by design
    g_create_err = std::string("hipGetDeviceCount: ") + (e != hipSuccess ? hipGetErrorString(e) : "no device");
    return GHF_E_HIP;
  }
  if (device < 0 || device >= ndev) return GHF_E_INVAL;
  ghf_ctx* c = new (std::nothrow) ghf_ctx();
  if (!c) return GHF_E_NOMEM;
  c->device = device;
  const char* what = "hipSetDevice";
  e = hipSetDevice(device);
#define GHF_STEP(call)      \
  if (e == hipSuccess) {    \
    what = #call;           \
    e = (call);             \
  }
  GHF_STEP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  GHF_STEP(hipMalloc(&c->d_status, sizeof(int)));
  GHF_STEP(hipHostMalloc(&c->h_status, sizeof(int), hipHostMallocDefault));
  GHF_STEP(hipMalloc(&c->d_hist, GHF_NSYM * sizeof(uint64_t)));
  GHF_STEP(hipMalloc(&c->d_hist_acc, kHistAccWords * sizeof(uint64_t)));
  GHF_STEP(hipMemset(c->d_hist_acc, 0, kHistAccWords * sizeof(uint64_t)));
  GHF_STEP(hipMalloc(&c->d_code, sizeof(ghf_code)));
  GHF_STEP(hipMalloc(&c->d_tree, sizeof(ghf_tree)));
  GHF_STEP(hipMalloc(&c->d_dt, sizeof(DecTables)));
  GHF_STEP(hipMalloc(&c->d_u64, 16 * sizeof(uint64_t)));
  GHF_STEP(hipHostMalloc(&c->h_u64, 16 * sizeof(uint64_t), hipHostMallocDefault));
  GHF_STEP(hipMemset(c->d_status, 0, sizeof(int)));
#undef GHF_STEP
  if (e != hipSuccess) {
    g_create_err = std::string(what) + ": " + hipGetErrorString(e);
    ghf_ctx_destroy(c);
    return GHF_E_HIP;
  }
  c->stream = c->own_stream;
  *out = c;
  return GHF_OK;
}

int ghf_ctx_destroy(ghf_ctx* c) {
  if (!c) return GHF_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->d_status) (void)hipFree(c->d_status);
  if (c->h_status) (void)hipHostFree(c->h_status);
  if (c->d_chunk_hist) (void)hipFree(c->d_chunk_hist);
  if (c->d_chunk_off) (void)hipFree(c->d_chunk_off);
  if (c->d_hist) (void)hipFree(c->d_hist);
  if (c->d_hist_acc) (void)hipFree(c->d_hist_acc);
  if (c->d_code) (void)hipFree(c->d_code);
  if (c->d_tree) (void)hipFree(c->d_tree);
  if (c->d_dt) (void)hipFree(c->d_dt);
  if (c->d_u64) (void)hipFree(c->d_u64);
  if (c->d_totals) (void)hipFree(c->d_totals);
  if (c->h_u64) (void)hipHostFree(c->h_u64);
  if (c->d_sync) (void)hipFree(c->d_sync);
  if (c->d_seg_abs) (void)hipFree(c->d_seg_abs);
  if (c->fidx.d_chunk_bit) (void)hipFree(c->fidx.d_chunk_bit);
  if (c->fidx.d_seg_bit) (void)hipFree(c->fidx.d_seg_bit);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  return GHF_OK;
}

int ghf_ctx_set_stream(ghf_ctx* c, void* hip_stream) {
  if (!c) return GHF_E_INVAL;
  c->stream = reinterpret_cast<hipStream_t>(hip_stream);  // NULL is HIP's default (null) stream
  return GHF_OK;
}

int ghf_sync(ghf_ctx* c) {
  if (!c) return GHF_E_INVAL;
  GHF_HIP(c, hipSetDevice(c->device));
  GHF_HIP(c, hipMemcpyAsync(c->h_status, c->d_status, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  GHF_HIP(c, hipStreamSynchronize(c->stream));
  const int s = *c->h_status;
  if (s != GHF_OK) c->err = ghf_status_string(s);
  return s;
}

int ghf_status(ghf_ctx* c) { return ghf_sync(c); }

int ghf_clear_status(ghf_ctx* c) {
  if (!c) return GHF_E_INVAL;
  GHF_HIP(c, hipSetDevice(c->device));
  GHF_HIP(c, hipMemsetAsync(c->d_status, 0, sizeof(int), c->stream));
  GHF_HIP(c, hipMemsetAsync(c->d_hist_acc, 0, kHistAccWords * sizeof(uint64_t), c->stream));  // in case a launch died half-way
  c->err.clear();
  return GHF_OK;
}

const char* ghf_last_error(ghf_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

// ---------------------------------------------------------------------------------------------- memory
int ghf_device_alloc(ghf_ctx* c, size_t bytes, void** d_ptr) {
  if (!c || !d_ptr) return GHF_E_INVAL;
  GHF_HIP(c, hipSetDevice(c->device));
  GHF_HIP(c, hipMalloc(d_ptr, bytes ? bytes : 16));
  return GHF_OK;
}
int ghf_device_free(ghf_ctx* c, void* d_ptr) {
  if (!c) return GHF_E_INVAL;
  if (d_ptr) {
    GHF_HIP(c, hipSetDevice(c->device));
    GHF_HIP(c, hipStreamSynchronize(c->stream));
    GHF_HIP(c, hipFree(d_ptr));
  }
  return GHF_OK;
}
int ghf_host_alloc(ghf_ctx* c, size_t bytes, void** h_ptr) {
  if (!c || !h_ptr) return GHF_E_INVAL;
  GHF_HIP(c, hipSetDevice(c->device));
  GHF_HIP(c, hipHostMalloc(h_ptr, bytes ? bytes : 16, hipHostMallocDefault));
  return GHF_OK;
}
int ghf_host_free(ghf_ctx* c, void* h_ptr) {
  if (!c) return GHF_E_INVAL;
  if (h_ptr) GHF_HIP(c, hipHostFree(h_ptr));
  return GHF_OK;
}
int ghf_copy_h2d(ghf_ctx* c, void* d_dst, const void* h_src, size_t bytes) {
  if (!c || (bytes && (!d_dst || !h_src))) return GHF_E_INVAL;
  GHF_HIP(c, hipSetDevice(c->device));
  if (bytes) GHF_HIP(c, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
  return GHF_OK;
}
int ghf_copy_d2h(ghf_ctx* c, void* h_dst, const void* d_src, size_t bytes) {
  if (!c || (bytes && (!h_dst || !d_src))) return GHF_E_INVAL;
  GHF_HIP(c, hipSetDevice(c->device));
  if (bytes) GHF_HIP(c, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
  return GHF_OK;
}
int ghf_copy_d2d(ghf_ctx* c, void* d_dst, const void* d_src, size_t bytes, int non_temporal) {
  if (!c || (bytes && (!d_dst || !d_src))) return GHF_E_INVAL;
  if (!aligned16(d_dst) || !aligned16(d_src)) return fail(c, GHF_E_INVAL, "ghf_copy_d2d: 16-byte aligned pointers");
  GHF_HIP(c, hipSetDevice(c->device));
  if (bytes) launch_stream_copy(static_cast<const uint8_t*>(d_src), static_cast<uint8_t*>(d_dst), bytes, non_temporal != 0, c->stream);
  return GHF_OK;
}
int ghf_memset_d(ghf_ctx* c, void* d_dst, int value, size_t bytes) {
  if (!c || (bytes && !d_dst)) return GHF_E_INVAL;
  GHF_HIP(c, hipSetDevice(c->device));
  if (bytes) GHF_HIP(c, hipMemsetAsync(d_dst, value, bytes, c->stream));
  return GHF_OK;
}

// ---------------------------------------------------------------------------------------------- events
struct ghf_event {
  int device = 0;
  hipEvent_t ev = nullptr;
};
int ghf_event_create(ghf_ctx* c, ghf_event** out) {
  if (!c || !out) return GHF_E_INVAL;
  GHF_HIP(c, hipSetDevice(c->device));
  ghf_event* e = new (std::nothrow) ghf_event;
  if (!e) return fail(c, GHF_E_HIP, "out of host memory");
  e->device = c->device;
  const hipError_t rc = hipEventCreateWithFlags(&e->ev, hipEventDisableTiming);
  if (rc != hipSuccess) {
    delete e;
    return fail(c, GHF_E_HIP, "hipEventCreateWithFlags", rc);
  }
  *out = e;
  return GHF_OK;
}
int ghf_event_destroy(ghf_event* e) {
  if (!e) return GHF_OK;
  (void)hipSetDevice(e->device);
  if (e->ev) (void)hipEventDestroy(e->ev);
  delete e;
  return GHF_OK;
}
int ghf_event_record(ghf_ctx* c, ghf_event* e) {
  if (!c || !e) return GHF_E_INVAL;
  GHF_HIP(c, hipSetDevice(c->device));
  GHF_HIP(c, hipEventRecord(e->ev, c->stream));
  return GHF_OK;
}
int ghf_event_wait(ghf_ctx* c, ghf_event* e) {
  if (!c || !e) return GHF_E_INVAL;
  GHF_HIP(c, hipSetDevice(c->device));
  GHF_HIP(c, hipStreamWaitEvent(c->stream, e->ev, 0));
  return GHF_OK;
}
int ghf_event_sync(ghf_event* e) {
  if (!e) return GHF_E_INVAL;
  if (hipSetDevice(e->device) != hipSuccess) return GHF_E_HIP;
  return hipEventSynchronize(e->ev) == hipSuccess ? GHF_OK : GHF_E_HIP;
}

// ---------------------------------------------------------------------------------------------- encode
uint32_t ghf_chunk_symbols(size_t n) { return chunk_symbols_for(n); }
size_t ghf_header_bytes(int max_len) { return 1040 + 8 * (size_t)max_len; }

size_t ghf_compress_bound(size_t n) {
  // header (max_len <= 32) + 9 bits per symbol (a 257-symbol Huffman code never loses to the 9-bit
  // fixed-length code) + end mark, rounded up to whole 16-byte units + one spare unit
  const size_t bits = 9 * (n + 1);
  size_t b = 1040 + 8 * 32 + (bits + 7) / 8;
  return ((b + 15) & ~(size_t)15) + 16;
}

static int histogram(ghf_ctx* c, const uint8_t* d_in, size_t n, uint64_t* d_hist, bool add) {
  if (!c || !d_hist || (n && !d_in)) return GHF_E_INVAL;
  GHF_HIP(c, hipSetDevice(c->device));
  const uint32_t cl = chunk_symbols_for(n);
  const size_t nchunks = (size_t)chunk_count_for(n);
  int rc = ensure_ws(c, nchunks);
  if (rc) return rc;
  launch_histogram(d_in, n, cl, (uint32_t)nchunks, c->d_chunk_hist, d_hist, c->d_hist_acc, add, c->stream);
  GHF_HIP(c, hipGetLastError());
  c->hist_in = add ? nullptr : d_in;  // a piece's buffer is refilled before anything is planned: nothing to keep
  c->hist_n = n;
  c->hist_chunk = cl;
  return GHF_OK;
}
int ghf_histogram(ghf_ctx* c, const uint8_t* d_in, size_t n, uint64_t* d_hist) { return histogram(c, d_in, n, d_hist, false); }
int ghf_histogram_add(ghf_ctx* c, const uint8_t* d_in, size_t n, uint64_t* d_hist) { return histogram(c, d_in, n, d_hist, true); }

int ghf_build_code_ex(ghf_ctx* c, const uint64_t* d_hist, ghf_code* d_code, unsigned flags) {
  if (!c || !d_hist || !d_code || (flags & ~(unsigned)(GHF_CODE_LIMIT | GHF_EMPTY_OK))) return GHF_E_INVAL;
  GHF_HIP(c, hipSetDevice(c->device));
  launch_build_code(d_hist, d_code, c->d_status, flags, c->stream);
  GHF_HIP(c, hipGetLastError());
  if (c->plan_code == d_code) c->plan_in = nullptr;  // tables changed: any cached plan is stale
  if (c->dt_code == d_code) c->dt_code = nullptr;
  return GHF_OK;
}

int ghf_build_code(ghf_ctx* c, const uint64_t* d_hist, ghf_code* d_code) { return ghf_build_code_ex(c, d_hist, d_code, 0); }

int ghf_write_header(ghf_ctx* c, const ghf_code* d_code, uint8_t* d_out, size_t cap) {
  if (!c || !d_code || !d_out || (reinterpret_cast<uintptr_t>(d_out) & 3u)) return GHF_E_INVAL;
  GHF_HIP(c, hipSetDevice(c->device));
  launch_write_header(d_code, d_out, cap, c->d_status, c->stream);
  GHF_HIP(c, hipGetLastError());
  return GHF_OK;
}

int ghf_encode_plan(ghf_ctx* c, const uint8_t* d_in, size_t n, const ghf_code* d_code, uint64_t* d_total_bits) {
  if (!c || !d_code || (n && !d_in)) return GHF_E_INVAL;
  GHF_HIP(c, hipSetDevice(c->device));
  const uint32_t cl = chunk_symbols_for(n);
  const size_t nchunks = (size_t)chunk_count_for(n);
  int rc = ensure_ws(c, nchunks);
  if (rc) return rc;
  const bool have_hist = c->hist_in == d_in && c->hist_n == n && c->hist_chunk == cl && n != 0;
  launch_plan(d_in, n, cl, (uint32_t)nchunks, have_hist ? c->d_chunk_hist : nullptr, d_code, c->d_chunk_off,
              d_total_bits ? d_total_bits : c->d_u64, c->stream);
  GHF_HIP(c, hipGetLastError());
  c->plan_in = d_in;
  c->plan_n = n;
  c->plan_code = d_code;
  return GHF_OK;
}

int ghf_encode_emit(ghf_ctx* c, const uint8_t* d_in, size_t n, const ghf_code* d_code, const uint64_t* d_start_bit,
                    int flags, uint8_t* d_out, size_t cap, const ghf_index* index, uint64_t* d_end) {
  if (!c || !d_code || !d_out || (n && !d_in)) return GHF_E_INVAL;
  if (!aligned16(d_out)) return fail(c, GHF_E_INVAL, "d_out must be 16-byte aligned");
  if ((flags & GHF_EMIT_HEADER) && (flags & GHF_EMIT_REBASE))
    return fail(c, GHF_E_INVAL, "GHF_EMIT_HEADER is for the buffer that starts at stream byte 0 (rank 0 / single GPU)");
  if (c->plan_in != d_in || c->plan_n != n || c->plan_code != d_code)
    return fail(c, GHF_E_INVAL, "ghf_encode_emit: call ghf_encode_plan on the same (d_in, n, d_code) first");
  GHF_HIP(c, hipSetDevice(c->device));
  const uint32_t cl = chunk_symbols_for(n);
  const size_t nchunks = (size_t)chunk_count_for(n);
  if (index) {
    if (index->n_symbols != n || index->chunk_symbols != (uint32_t)kBlockSymbols || index->seg_symbols != (uint32_t)kSegSymbols ||
        !index->d_chunk_bit || !index->d_seg_bit)
      return fail(c, GHF_E_INVAL, "ghf_encode_emit: index does not match n (use ghf_index_alloc)");
  }
  EmitParams p;
  p.in = d_in;
  p.n = n;
  p.code = d_code;
  p.chunk_off = c->d_chunk_off;
  p.d_start_bit = d_start_bit;
  p.out = d_out;
  p.cap = cap;
  p.chunk = cl;
  p.nchunks = (uint32_t)nchunks;
  p.chunk_bit = index ? index->d_chunk_bit : nullptr;
  p.seg_bit = index ? index->d_seg_bit : nullptr;
  p.flags = flags;
  p.status = c->d_status;
  p.d_end = d_end;
  launch_emit(p, c->stream);
  GHF_HIP(c, hipGetLastError());
  return GHF_OK;
}

int ghf_shard_start_bit(ghf_ctx* c, const ghf_code* d_code, const uint64_t* d_totals, int world, int rank,
                        uint64_t* d_start_bit) {
  if (!c || !d_code || !d_totals || !d_start_bit || world < 1 || rank < 0 || rank >= world) return GHF_E_INVAL;
  GHF_HIP(c, hipSetDevice(c->device));
  launch_shard_start(d_code, d_totals, rank, d_start_bit, c->stream);
  GHF_HIP(c, hipGetLastError());
  return GHF_OK;
}

int ghf_compress(ghf_ctx* c, const uint8_t* d_in, size_t n, uint8_t* d_out, size_t cap, uint64_t* d_out_bytes,
                 ghf_code* d_code, const ghf_index* index) {
  return ghf_compress_ex(c, d_in, n, d_out, cap, d_out_bytes, d_code, index, 0);
}

int ghf_compress_ex(ghf_ctx* c, const uint8_t* d_in, size_t n, uint8_t* d_out, size_t cap, uint64_t* d_out_bytes,
                    ghf_code* d_code, const ghf_index* index, unsigned code_flags) {
  if (!c || !d_out || (n && !d_in)) return GHF_E_INVAL;
  if (n == 0 && !(code_flags & GHF_EMPTY_OK)) return fail(c, GHF_E_EMPTY, "empty input is undefined in the reference; refused");
  if (!aligned16(d_out)) return fail(c, GHF_E_INVAL, "d_out must be 16-byte aligned");
  ghf_code* code = d_code ? d_code : c->d_code;
  int rc;
  if (n == 0) {
    // GHF_EMPTY_OK (include/ghf.h): header of the one-symbol code + the byte 0x7F.  No K4/K5: there is nothing to pack.
    const size_t hdr = ghf_header_bytes(1);
    if (cap < hdr + 1) return fail(c, GHF_E_CAP, "ghf_compress: capacity below the 1049 bytes of the empty stream");
    if ((rc = ghf_histogram(c, d_in, 0, c->d_hist))) return rc;
    if ((rc = ghf_build_code_ex(c, c->d_hist, code, code_flags))) return rc;
    if ((rc = ghf_write_header(c, code, d_out, cap))) return rc;
    GHF_HIP(c, hipMemsetAsync(d_out + hdr, 0x7F, 1, c->stream));
    if (d_out_bytes) {
      launch_store_u64(d_out_bytes, nullptr, hdr + 1, c->stream);
      GHF_HIP(c, hipGetLastError());
    }
    return GHF_OK;
  }
  if ((rc = ghf_histogram(c, d_in, n, c->d_hist))) return rc;   // compressor.h:63
  if ((rc = ghf_build_code_ex(c, c->d_hist, code, code_flags))) return rc;  // compressor.h:64
  if ((rc = ghf_encode_plan(c, d_in, n, code, c->d_u64))) return rc;
  // compressor.h:70 + :72 -- the header rides along with the emit launches
  if ((rc = ghf_encode_emit(c, d_in, n, code, nullptr, GHF_EMIT_LAST | GHF_EMIT_HEADER, d_out, cap, index, c->d_u64 + 1))) return rc;
  if (d_out_bytes) {
    launch_store_u64(d_out_bytes, c->d_u64 + 2, 0, c->stream);
    GHF_HIP(c, hipGetLastError());
  }
  return GHF_OK;
}

// ---------------------------------------------------------------------------------------------- index
int ghf_index_alloc(ghf_ctx* c, size_t n_symbols, ghf_index* out) {
  if (!c || !out) return GHF_E_INVAL;
  GHF_HIP(c, hipSetDevice(c->device));
  std::memset(out, 0, sizeof *out);
  out->n_symbols = n_symbols;
  out->chunk_symbols = kBlockSymbols;
  out->seg_symbols = kSegSymbols;
  out->n_chunks = (n_symbols + kBlockSymbols - 1) / kBlockSymbols;
  out->n_segs = (n_symbols + kSegSymbols - 1) / kSegSymbols;
  GHF_HIP(c, hipMalloc(&out->d_chunk_bit, std::max<size_t>(out->n_chunks, 1) * sizeof(uint64_t)));
  hipError_t e = hipMalloc(&out->d_seg_bit, std::max<size_t>(out->n_segs, 1) * sizeof(uint32_t));
  if (e != hipSuccess) {
    (void)hipFree(out->d_chunk_bit);
    out->d_chunk_bit = nullptr;
    return fail(c, GHF_E_HIP, "hipMalloc(seg_bit)", e);
  }
  return GHF_OK;
}

int ghf_index_free(ghf_ctx* c, ghf_index* idx) {
  if (!c || !idx) return GHF_E_INVAL;
  GHF_HIP(c, hipSetDevice(c->device));
  GHF_HIP(c, hipStreamSynchronize(c->stream));
  if (idx->d_chunk_bit) (void)hipFree(idx->d_chunk_bit);
  if (idx->d_seg_bit) (void)hipFree(idx->d_seg_bit);
  idx->d_chunk_bit = nullptr;
  idx->d_seg_bit = nullptr;
  return GHF_OK;
}

// ---------------------------------------------------------------------------------------------- decode
static inline uint32_t be32(const uint8_t* p) {
  return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

// canonical_huff_encoder.cc:349-374, plus the validation the reference does not do.
int ghf_parse_header(const uint8_t* h, size_t n, ghf_code* code, size_t* header_bytes) {
  if (!h || !code) return GHF_E_INVAL;
  if (n < 1040) return GHF_E_FORMAT;
  if (be32(h) != GHF_NSYM) return GHF_E_FORMAT;
  std::memset(code, 0, sizeof *code);
  const uint8_t* p = h + 4;
  for (int i = 0; i < GHF_NSYM; ++i, p += 4) code->symbol[i] = be32(p);
  const uint32_t min_len = be32(p), max_len = be32(p + 4);
  p += 8;
  if (max_len < 1 || max_len > 32 || min_len < 1 || min_len > max_len) return GHF_E_FORMAT;
  if (n < 1040 + 8 * (size_t)max_len) return GHF_E_FORMAT;
  code->min_len = (int32_t)min_len;
  code->max_len = (int32_t)max_len;
  for (uint32_t i = 1; i <= max_len; ++i, p += 8) {
    code->start_pos[i] = be32(p);
    code->first_code[i] = be32(p + 4);
  }
  // used symbols are a prefix of symbol_[], all distinct, the end mark among them
  uint32_t used = 0;
  while (used < GHF_NSYM && code->symbol[used] != 0xFFFFFFFFu) ++used;
  bool seen[GHF_NSYM] = {false};
  for (uint32_t i = 0; i < GHF_NSYM; ++i) {
    const uint32_t s = code->symbol[i];
    if (i < used) {
      if (s >= GHF_NSYM || seen[s]) return GHF_E_FORMAT;
      seen[s] = true;
    } else if (s != 0xFFFFFFFFu) {
      return GHF_E_FORMAT;
    }
  }
  if (!seen[GHF_NSYM - 1]) return GHF_E_FORMAT;
  // one symbol only: the empty stream of GHF_EMPTY_OK -- the end mark alone, code "0" (not a complete code: checked here, not below)
  if (used == 1) {
    if (max_len != 1 || code->first_code[1] != 0 || code->start_pos[1] != 0) return GHF_E_FORMAT;
    code->length[GHF_NSYM - 1] = 1;
    code->codeword[GHF_NSYM - 1] = 0;
    if (header_bytes) *header_bytes = 1040 + 8;
    return GHF_OK;
  }
  // rebuild per-symbol lengths/codewords; the code must be the canonical complete prefix code
  uint64_t kraft = 0;  // in units of 2^-32
  for (uint32_t len = min_len; len <= max_len; ++len) {
    const uint32_t a = code->start_pos[len];
    const uint32_t b = (len < max_len) ? code->start_pos[len + 1] : used;
    if (a > b || b > used) return GHF_E_FORMAT;
    if (len == min_len && a != 0) return GHF_E_FORMAT;
    const uint64_t fc = code->first_code[len];
    if (fc + (b - a) > (1ull << len)) return GHF_E_FORMAT;
    for (uint32_t r = 0; r < b - a; ++r) {
      const uint32_t s = code->symbol[a + r];
      code->length[s] = len;
      code->codeword[s] = (uint32_t)fc + r;
    }
    kraft += (uint64_t)(b - a) << (32 - len);
    if (len < max_len) {
      // canonical_huff_encoder.cc:109-114: first_code[l] = (first_code[l+1] + num[l+1]) / 2
      const uint32_t nb = ((len + 1 < max_len) ? code->start_pos[len + 2] : used) - code->start_pos[len + 1];
      if (fc != ((uint64_t)code->first_code[len + 1] + nb) / 2) return GHF_E_FORMAT;
    } else if (fc != 0) {
      return GHF_E_FORMAT;
    }
  }
  for (uint32_t i = 1; i < min_len; ++i)
    if (code->first_code[i] != 1024) return GHF_E_FORMAT;  // canonical_huff_encoder.cc:119-121
  if (kraft != (1ull << 32)) return GHF_E_FORMAT;
  if (header_bytes) *header_bytes = 1040 + 8 * (size_t)max_len;
  return GHF_OK;
}

// K6: rebuild the side-car of a stream that came without one (e.g. a .crs2 written by the reference).
// Synchronises with the host a few times (convergence flag, symbol count); fills c->fidx.
static int rebuild_index_at(ghf_ctx* c, const uint8_t* d_stream, size_t stream_bytes, size_t hdr, uint64_t end_bit, int mode,
                            size_t cap, uint32_t first_start = 0, uint64_t* landing = nullptr, int* has_end_mark = nullptr,
                            bool prefer_scan = false, int max_len_hint = 0);

static int rebuild_index(ghf_ctx* c, const uint8_t* d_stream, size_t stream_bytes, const ghf_code* d_code, size_t cap) {
  ghf_code* hc = new (std::nothrow) ghf_code;
  if (!hc) return GHF_E_NOMEM;
  hipError_t e = hipMemcpyAsync(hc, d_code, sizeof(ghf_code), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  const int max_len = hc->max_len, min_len = hc->min_len;
  delete hc;
  if (e != hipSuccess) return fail(c, GHF_E_HIP, "copy tables to host", e);
  if (max_len < 1 || max_len > 32) return fail(c, GHF_E_FORMAT, "bad max_len in tables");
  const size_t hdr = ghf_header_bytes(max_len);
  if (stream_bytes <= hdr) return fail(c, GHF_E_FORMAT, "stream shorter than its header");
  return rebuild_index_at(c, d_stream, stream_bytes, hdr, (uint64_t)stream_bytes * 8, 0, cap, 0, nullptr, nullptr,
                          /*prefer_scan=*/max_len - min_len <= 1, max_len);
}

// K6 driver.  hdr = bytes in front of the first code; end_bit = one past the last bit that may belong to a code;
// mode 0: .crs2 (ends with the end mark); 1: .crs (no end mark, must end exactly at end_bit); 2: a piece of a .crs2
// whose first code boundary is assumed first_start bits behind hdr and whose last code may run past end_bit
static int rebuild_index_at(ghf_ctx* c, const uint8_t* d_stream, size_t stream_bytes, size_t hdr, uint64_t end_bit, int mode,
                            size_t cap, uint32_t first_start, uint64_t* landing, int* has_end_mark, bool prefer_scan, int max_len_hint) {
  const bool no_eof = mode == 1;
  SyncParams p;
  p.stream = d_stream;
  p.stream_bytes = stream_bytes;
  p.body_bit0 = (uint64_t)hdr * 8;
  p.end_bit = end_bit;
  p.no_eof = (uint32_t)mode;
  p.dt = c->d_dt;
  p.nsub = (end_bit - p.body_bit0 + 511) / 512;
  const size_t ntiles = (p.nsub + 255) / 256;
  // carve the workspace
  size_t off = 0;
  auto carve = [&](size_t bytes) {
    const size_t o = off;
    off += (bytes + 255) & ~(size_t)255;
    return o;
  };
  const size_t o_start = carve((p.nsub + 1) * 2), o_used = carve(p.nsub * 2), o_cnt = carve(p.nsub * 4), o_eof = carve(p.nsub),
               o_tile = carve((ntiles + 2) * 8), o_scan = carve(sync_scan_workspace(p.nsub));
  if (off > c->sync_cap) {
    if (c->d_sync) (void)hipFree(c->d_sync);
    c->d_sync = nullptr;
    c->sync_cap = 0;
    GHF_HIP(c, hipMalloc(&c->d_sync, off));
    c->sync_cap = off;
  }
  uint8_t* ws = static_cast<uint8_t*>(c->d_sync);
  p.start = reinterpret_cast<uint16_t*>(ws + o_start);
  p.used = reinterpret_cast<uint16_t*>(ws + o_used);
  p.cnt = reinterpret_cast<uint32_t*>(ws + o_cnt);
  p.eof = ws + o_eof;
  p.tile_sum = reinterpret_cast<uint64_t*>(ws + o_tile);
  p.changed = reinterpret_cast<uint32_t*>(c->d_u64 + 9);
  p.moved_first_inv = reinterpret_cast<unsigned long long*>(c->d_u64 + 10);
  p.eof_sub = c->d_u64 + 4;
  // `start`, `used`, `eof` (11 bytes per KiB of stream: 3 + 3 + 1.5 MB at 256 MiB) are not cleared when the fixed-point passes
  // come first: the first k_sync_pass of the call finds every subsequence unworked and every guess 0 because its parameters
  // say so (SyncParams::first), and writes all three arrays whole.  (Round 3 queued three memsets, 60 us at 256 MiB in front
  // of a 1 ms job.)
  const bool scan_first = prefer_scan && first_start < ((max_len_hint >= 1 && max_len_hint <= 16) ? 16u : (max_len_hint > 32 ? 64u : 32u));
  if (scan_first) {
    // (the scan's kernels write the guesses they derive -- and, for the subsequences whose class walk already is the whole
    //  answer, the results too: `used` must say "nothing yet" for all the others)
    GHF_HIP(c, hipMemsetAsync(p.start, 0, (p.nsub + 1) * 2, c->stream));
    if (first_start) launch_store_u64(reinterpret_cast<uint64_t*>(p.start), nullptr, first_start, c->stream);  // start[0]
    GHF_HIP(c, hipMemsetAsync(p.used, 0xFF, p.nsub * 2, c->stream));
    GHF_HIP(c, hipMemsetAsync(p.eof, 0, p.nsub, c->stream));
  } else {
    launch_store_u64(reinterpret_cast<uint64_t*>(p.start), nullptr, first_start, c->stream);  // start[0] (k_sync_pass stores start[1 ..])
  }
  p.first = scan_first ? 0u : 3u;
  p.first_start = first_start;
  // passes until no boundary guess moves (self-synchronisation: a handful of passes in practice).  They are queued in
  // batches; behind every batch the counts (first end mark, symbols per tile, their scan) and the landing bit are queued as
  // well, and the host reads {symbols, end-mark subsequence, "something moved", landing} in ONE round trip: a stream that has
  // settled in its first batch -- the usual case, and the rule behind the deterministic scan -- costs one synchronisation (round 2:
  // four per call, a quarter of the file decompressor's K6 time at 16 MiB pieces).
  // Streams that self-synchronise slowly (near-fixed-length codes) would need one pass per subsequence of drift: their
  // boundaries are seeded by the deterministic scan (launch_sync_scan) -- at once when the caller knows the code is of that
  // kind, otherwise as soon as a first batch of passes has not settled.  The passes then only verify.
  uint64_t n = 0, eof_sub = 0;
  uint16_t land16 = 0;
  {
    constexpr int kBatch = 4;
    uint64_t passes = 0;
    const uint32_t fn_stride = (max_len_hint >= 1 && max_len_hint <= 16) ? 16u : (max_len_hint > 32 ? 64u : 32u);
    bool scanned = first_start >= fn_stride;  // (the scan follows entry offsets below its stride only)
    if (prefer_scan && !scanned) {
      launch_sync_scan(p, ws + o_scan, fn_stride, first_start, c->stream);
      scanned = true;
    }
    for (;;) {
      if (passes > p.nsub + 2) return fail(c, GHF_E_CORRUPT, "self-synchronisation did not converge");
      // (behind the deterministic scan the first pass only has to CONFIRM the boundaries: one pass, not a batch)
      const int nb = (scanned && passes == 0) ? 1 : kBatch;
      for (int b = 0; b < nb; ++b) {
        GHF_HIP(c, hipMemsetAsync(p.changed, 0, 16, c->stream));  // flag, count, first (inverted)
        launch_sync_pass(p, c->stream);
        p.first = 0;
      }
      passes += nb;
      launch_store_u64(p.eof_sub, nullptr, p.nsub, c->stream);  // "none found"; k_sync_eof takes the minimum
      launch_sync_counts(p, c->d_u64 + 3, c->stream);
      launch_load_u16(c->d_u64 + 8, p.start + p.nsub, c->stream);
      GHF_HIP(c, hipMemcpyAsync(c->h_u64 + 3, c->d_u64 + 3, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
      GHF_HIP(c, hipStreamSynchronize(c->stream));
#ifdef GHF_K6_TRACE  // (scratch/build_variant.sh k6trace -DGHF_K6_TRACE: what every batch of passes left behind)
      fprintf(stderr, "[k6] nsub %llu passes %llu scanned %d flag %u moved~%llu first %llu eof_sub %llu\n", (unsigned long long)p.nsub,
              (unsigned long long)passes, (int)scanned, (unsigned)c->h_u64[9], (unsigned long long)((c->h_u64[9] >> 32) * 256),
              (unsigned long long)~c->h_u64[10], (unsigned long long)c->h_u64[4]);
#endif
      if ((uint32_t)c->h_u64[9] == 0) break;
      // .crs2: settled IN FRONT OF THE END MARK is settled.  No landing at or before the first end mark's subsequence moved
      // in the batch's last pass -> every boundary up to it is a fixed point, the mark is the stream's (the first one, and
      // real).  What lies behind it (a buffer longer than its stream: stale bytes) may go on moving for ever.
      if (!no_eof && mode != 2 && c->h_u64[10] != 0 && ~c->h_u64[10] >= c->h_u64[4] && c->h_u64[4] < p.nsub) break;
      const uint64_t moved = (c->h_u64[9] >> 32) * 256;  // boundaries the batch's last pass moved (sampled: every 256th group)
      // Not settled.  A boundary that is still moving travels ONE subsequence to the right per pass, and a pass in which
      // little moved costs little (a wave whose 64 subsequences are current skips them), so a few more batches are cheaper
      // than the deterministic scan over the whole stream -- which long streams of a quickly synchronising code otherwise
      // fall into because SOME stretch among their millions of subsequences needs a fifth pass (4 GiB Zipf: 44 ms with
      // the scan after the first batch, see profiles/r04/foreign_4GiB.txt).  Codes that do not settle in kScanAfter passes
      // (long runs of one value, near-fixed-length codes the caller did not announce) get the scan then.
      // What decides is HOW MUCH still moves: a few stragglers (one stretch in a 4 GiB Zipf stream needs 17..20 passes,
      // whichever seed) are followed for up to kScanAfterFew passes; a stream in which boundaries still move everywhere
      // after two batches is not going to settle by itself.
      constexpr uint64_t kScanAfterMany = 8, kScanAfterFew = 64;
      const uint64_t few = p.nsub >> 10 > 4096 ? p.nsub >> 10 : 4096;
      if (!scanned && (passes >= kScanAfterFew || (passes >= kScanAfterMany && moved > few))) {
        launch_sync_scan(p, ws + o_scan, fn_stride, first_start, c->stream);
        scanned = true;
      }
    }
    n = c->h_u64[3];
    eof_sub = c->h_u64[4];
    land16 = (uint16_t)c->h_u64[8];
  }
  if (no_eof) {
    // every subsequence counts; a flagged one means bits that are no code or a code running past the end
    // (eof_sub == nsub, "none", makes the counting kernels take every subsequence: n is already the total)
    if (eof_sub < p.nsub) return fail(c, GHF_E_CORRUPT, "the .crs body does not end on a code boundary");
  } else if (mode == 2) {
    if (has_end_mark) *has_end_mark = eof_sub < p.nsub;
    if (landing) *landing = land16 == 0xFFFF ? 0 : land16;  // 0xFFFF: the last subsequence ended at an end mark (real, or a fake one of a wrong guess)
  } else if (eof_sub >= p.nsub) {
    return fail(c, GHF_E_CORRUPT, "no end mark in the stream");
  }
  if (n > cap) return fail(c, GHF_E_CAP, "ghf_decode: output capacity below the decoded size");
  // size the side-car
  ghf_index& ix = c->fidx;
  const uint64_t n_chunks = (n + kBlockSymbols - 1) / kBlockSymbols, n_segs = (n + kSegSymbols - 1) / kSegSymbols;
  if (n_segs > c->fidx_cap_segs) {
    if (ix.d_seg_bit) (void)hipFree(ix.d_seg_bit);
    if (c->d_seg_abs) (void)hipFree(c->d_seg_abs);
    ix.d_seg_bit = nullptr;
    c->d_seg_abs = nullptr;
    c->fidx_cap_segs = 0;
    GHF_HIP(c, hipMalloc(&ix.d_seg_bit, n_segs * sizeof(uint32_t)));
    GHF_HIP(c, hipMalloc(&c->d_seg_abs, (n_segs + 1) * sizeof(uint64_t)));
    c->fidx_cap_segs = n_segs;
  }
  if (n_chunks > c->fidx_cap_chunks) {
    if (ix.d_chunk_bit) (void)hipFree(ix.d_chunk_bit);
    ix.d_chunk_bit = nullptr;
    c->fidx_cap_chunks = 0;
    GHF_HIP(c, hipMalloc(&ix.d_chunk_bit, n_chunks * sizeof(uint64_t)));
    c->fidx_cap_chunks = n_chunks;
  }
  ix.n_symbols = n;
  ix.chunk_symbols = kBlockSymbols;
  ix.seg_symbols = kSegSymbols;
  ix.n_chunks = n_chunks;
  ix.n_segs = n_segs;
  ix.flags = (mode == 2 && eof_sub >= p.nsub) ? (uint32_t)GHF_INDEX_NO_END_MARK : 0u;
  if (n) launch_sync_index(p, c->d_seg_abs, n, ix.d_chunk_bit, ix.d_seg_bit, c->stream);
  GHF_HIP(c, hipGetLastError());
  c->fidx_stream = d_stream;
  c->fidx_bytes = stream_bytes;
  return GHF_OK;
}

// Multi-GPU decode of a side-car-less stream: one rank's piece.  Rebuilds the piece's side-car and keeps it
// for the following ghf_decode(index = NULL) of the same (d_piece, piece_bytes).
int ghf_sync_piece(ghf_ctx* c, const uint8_t* d_piece, size_t piece_bytes, uint32_t first_bit, uint64_t end_bit,
                   const ghf_code* d_code, uint64_t* landing, uint64_t* n_symbols, int* has_end_mark) {
  if (!c || !d_piece || !d_code || !landing || !n_symbols || !has_end_mark) return GHF_E_INVAL;
  if (!aligned16(d_piece)) return fail(c, GHF_E_INVAL, "d_piece must be 16-byte aligned");
  if (first_bit >= 512 || end_bit > (uint64_t)piece_bytes * 8 || first_bit > end_bit) return fail(c, GHF_E_INVAL, "ghf_sync_piece: bad first_bit / end_bit");
  GHF_HIP(c, hipSetDevice(c->device));
  launch_build_decode_tables(d_code, c->d_dt, c->d_status, c->stream);
  c->dt_code = nullptr;
  c->fidx_stream = nullptr;
  *landing = 0;
  *n_symbols = 0;
  *has_end_mark = 0;
  if (end_bit == 0) {  // nothing of this piece is its own
    c->fidx.n_symbols = 0;
    c->fidx.n_segs = 0;
    c->fidx.n_chunks = 0;
    c->fidx_stream = d_piece;
    c->fidx_bytes = piece_bytes;
    return GHF_OK;
  }
  // near-fixed-length codes re-synchronise slowly: the deterministic scan seeds the boundaries at once (as in rebuild_index)
  GHF_HIP(c, hipMemcpyAsync(c->h_u64 + 6, &d_code->min_len, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  GHF_HIP(c, hipStreamSynchronize(c->stream));
  int32_t lens[2];
  std::memcpy(lens, c->h_u64 + 6, sizeof lens);
  if (lens[1] < 1 || lens[1] > 32 || lens[0] < 1 || lens[0] > lens[1]) return fail(c, GHF_E_FORMAT, "bad min_len / max_len in tables");
  const int rc = rebuild_index_at(c, d_piece, piece_bytes, 0, end_bit, 2, (size_t)-1, first_bit, landing, has_end_mark,
                                  /*prefer_scan=*/lens[1] - lens[0] <= 1, lens[1]);
  if (rc) return rc;
  *n_symbols = c->fidx.n_symbols;
  return GHF_OK;
}

int ghf_decoded_size(ghf_ctx* c, const uint8_t* d_stream, size_t stream_bytes, const ghf_code* d_code, uint64_t* n_out) {
  if (!c || !d_stream || !d_code || !n_out) return GHF_E_INVAL;
  if (!aligned16(d_stream)) return fail(c, GHF_E_INVAL, "d_stream must be 16-byte aligned");
  GHF_HIP(c, hipSetDevice(c->device));
  launch_build_decode_tables(d_code, c->d_dt, c->d_status, c->stream);
  c->dt_code = nullptr;
  c->fidx_stream = nullptr;
  const int rc = rebuild_index(c, d_stream, stream_bytes, d_code, (size_t)-1);
  if (rc) return rc;
  *n_out = c->fidx.n_symbols;
  return GHF_OK;
}

int ghf_decode_prepare(ghf_ctx* c, const ghf_code* d_code) {
  if (!c || !d_code) return GHF_E_INVAL;
  GHF_HIP(c, hipSetDevice(c->device));
  launch_build_decode_tables(d_code, c->d_dt, c->d_status, c->stream);
  GHF_HIP(c, hipGetLastError());
  c->dt_code = d_code;
  return GHF_OK;
}

int ghf_decode(ghf_ctx* c, const uint8_t* d_stream, size_t stream_bytes, const ghf_code* d_code,
               const ghf_index* index, uint8_t* d_out, size_t cap, uint64_t* d_out_bytes) {
  if (!c || !d_stream || !d_code || !d_out) return GHF_E_INVAL;
  if (!aligned16(d_stream)) return fail(c, GHF_E_INVAL, "d_stream must be 16-byte aligned");
  GHF_HIP(c, hipSetDevice(c->device));
  if (c->dt_code == d_code && index) c->dt_code = nullptr;  // prepared: single use (the tables also hold this decode's work counters)
  else {
    c->dt_code = nullptr;
    launch_build_decode_tables(d_code, c->d_dt, c->d_status, c->stream);
  }
  if (!index) {
    if (c->fidx_stream != d_stream || c->fidx_bytes != stream_bytes) {  // else: ghf_decoded_size already did it
      const int rc = rebuild_index(c, d_stream, stream_bytes, d_code, cap);
      if (rc) return rc;
    }
    c->fidx_stream = nullptr;  // single use: the buffer may be rewritten afterwards
    index = &c->fidx;
  }
  if (index->n_symbols == 0) {  // the end mark comes first (GHF_EMPTY_OK's stream, or a shard that holds nothing): no K7
    if (d_out_bytes) {
      launch_store_u64(d_out_bytes, nullptr, 0, c->stream);
      GHF_HIP(c, hipGetLastError());
    }
    return GHF_OK;
  }
  if (!index->d_chunk_bit || !index->d_seg_bit || index->seg_symbols != (uint32_t)kSegSymbols ||
      index->chunk_symbols != (uint32_t)kBlockSymbols || index->n_segs != (index->n_symbols + kSegSymbols - 1) / kSegSymbols ||
      index->n_chunks != (index->n_symbols + kBlockSymbols - 1) / kBlockSymbols)
    return fail(c, GHF_E_INVAL, "ghf_decode: malformed index");
  if (cap < index->n_symbols) return fail(c, GHF_E_CAP, "ghf_decode: output capacity below n_symbols");
  if (index->n_chunks >= kDecMaxGroups) return fail(c, GHF_E_INVAL, "ghf_decode: more than 2^44 symbols in one call");
  DecParams p;
  p.stream = d_stream;
  p.stream_bytes = stream_bytes;
  p.dt = c->d_dt;
  p.chunk_bit = index->d_chunk_bit;
  p.seg_bit = index->d_seg_bit;
  p.n_symbols = index->n_symbols;
  p.n_segs = index->n_segs;
  p.no_end_mark = (index->flags & GHF_INDEX_NO_END_MARK) ? 1u : 0u;
  p.out = d_out;
  p.status = c->d_status;
  p.out_bytes = d_out_bytes;
  launch_decode(p, c->stream);
  if (d_out_bytes && index->n_symbols == 0) launch_store_u64(d_out_bytes, nullptr, 0, c->stream);  // no decode launch then
  GHF_HIP(c, hipGetLastError());
  return GHF_OK;
}

// ---------------------------------------------------------------------------------------------- .crs
int ghf_crs_build_code(ghf_ctx* c, const uint64_t* d_hist, ghf_tree* d_tree, ghf_code* d_code) {
  if (!c || !d_hist || !d_tree || !d_code) return GHF_E_INVAL;
  GHF_HIP(c, hipSetDevice(c->device));
  launch_crs_build_code(d_hist, d_tree, d_code, c->d_u64 + 6, c->d_status, c->stream);
  GHF_HIP(c, hipGetLastError());
  c->plan_in = nullptr;  // d_code changed: a cached plan no longer describes it
  if (c->dt_code == d_code) c->dt_code = nullptr;  // ... and neither do decode tables prepared from it
  return GHF_OK;
}

size_t ghf_crs_compress_bound(size_t n) { return 1024 + 2 + ghf_compress_bound(n); }

int ghf_crs_compress(ghf_ctx* c, const uint8_t* d_in, size_t n, uint8_t* d_out, size_t cap, uint64_t* d_out_bytes,
                     ghf_tree* d_tree, const ghf_index* index) {
  if (!c || !d_out || (n && !d_in)) return GHF_E_INVAL;
  if (n == 0) return fail(c, GHF_E_EMPTY, "empty input is undefined in the reference; refused");
  if (!aligned16(d_out)) return fail(c, GHF_E_INVAL, "d_out must be 16-byte aligned");
  ghf_tree* tree = d_tree ? d_tree : c->d_tree;
  int rc;
  if ((rc = ghf_histogram(c, d_in, n, c->d_hist))) return rc;            // compressor.h:63
  if ((rc = ghf_crs_build_code(c, c->d_hist, tree, c->d_code))) return rc;  // compressor.h:64, start bit -> d_u64[6]
  if ((rc = ghf_encode_plan(c, d_in, n, c->d_code, c->d_u64))) return rc;
  // compressor.h:72: the body right behind tree + two prefix bytes; no end mark, zero fill (flags = 0)
  if ((rc = ghf_encode_emit(c, d_in, n, c->d_code, c->d_u64 + 6, GHF_EMIT_LONG_CODES, d_out, cap, index, c->d_u64 + 1))) return rc;
  launch_crs_finish(tree, c->d_u64, d_out, d_out_bytes, c->d_status, c->stream);  // compressor.h:70 + normal_huff_encoder.h:176-184
  GHF_HIP(c, hipGetLastError());
  return GHF_OK;
}

// DecodeHuffTree::do_build_tree (include/huff_tree.cc:289-303), iteratively and with bounds
int ghf_crs_parse_header(const uint8_t* h, size_t n, ghf_tree* tree, size_t* tree_bytes) {
  if (!h || !tree) return GHF_E_INVAL;
  std::memset(tree, 0, sizeof *tree);
  struct Open {  // a parent that still waits for a child
    uint16_t idx, depth;
    bool has_left;
  };
  Open stack[260];
  int sp = 0;
  size_t pos = 0;
  uint32_t n_parents = 0, n_leaves = 0, max_len = 0;
  for (bool first_node = true;; first_node = false) {
    if (pos + 2 > n || pos + 2 > sizeof tree->header) return GHF_E_FORMAT;
    const bool leaf = h[pos] == 0;  // huff_tree.cc:294: first byte 0 = leaf, second byte = key
    const uint32_t key = h[pos + 1];
    pos += 2;
    uint32_t id, depth = 0;
    if (leaf) {
      id = key;
      ++n_leaves;
    } else {
      if (n_parents >= 255) return GHF_E_FORMAT;
      id = 256 + n_parents++;
    }
    if (first_node) {
      if (leaf) return GHF_E_FORMAT;  // the root is a leaf: the reference's decoder dereferences NULL there
      tree->root = id;
    } else {
      Open& top = stack[sp - 1];
      depth = top.depth + 1u;
      if (!top.has_left) {
        tree->left[top.idx] = (uint16_t)id;
        top.has_left = true;
      } else {
        tree->right[top.idx] = (uint16_t)id;
        --sp;
      }
    }
    if (leaf) {
      if (depth > max_len) max_len = depth;
    } else {
      if (sp >= 258) return GHF_E_FORMAT;
      stack[sp].idx = (uint16_t)(id - 256);
      stack[sp].depth = (uint16_t)depth;
      stack[sp].has_left = false;
      ++sp;
    }
    if (sp == 0) break;
  }
  if (n_leaves < 2 || n_leaves > 256 || n_parents != n_leaves - 1) return GHF_E_FORMAT;
  if (max_len > 64) return GHF_E_CODELEN;  // (a tree that deep needs more than 2^44 input bytes)
  tree->n_leaves = n_leaves;
  tree->max_len = max_len;
  tree->tree_bytes = (uint32_t)pos;
  std::memcpy(tree->header, h, pos);
  if (tree_bytes) *tree_bytes = pos;
  return GHF_OK;
}

// the depth of a device-resident tree (1..64), for the stride of K6's function rows: 16 / 32 / 64 bytes as for .crs2 -- a
// shallow tree does not pay for 64-byte rows
static int crs_max_len(ghf_ctx* c, const ghf_tree* d_tree, int* max_len) {
  hipError_t e = hipMemcpyAsync(c->h_u64 + 9, &d_tree->max_len, sizeof(d_tree->max_len), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(c, GHF_E_HIP, "copy the tree's max_len to host", e);
  uint32_t ml = 0;
  std::memcpy(&ml, c->h_u64 + 9, sizeof ml);
  *max_len = (ml >= 1 && ml <= 64) ? (int)ml : 64;  // (anything else is refused by k_crs_decode_tables)
  return GHF_OK;
}

static int crs_geometry(ghf_ctx* c, const ghf_tree* d_tree, size_t stream_bytes, int left_bits, size_t* hdr, uint64_t* end_bit, int* max_len) {
  if (left_bits < 0 || left_bits > 7) return fail(c, GHF_E_FORMAT, "left_bits must be 0..7");
  uint32_t tb = 0, ml = 0;
  hipError_t e = hipMemcpyAsync(c->h_u64 + 7, &d_tree->tree_bytes, 4, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(c->h_u64 + 9, &d_tree->max_len, sizeof(d_tree->max_len), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(c, GHF_E_HIP, "copy tree_bytes to host", e);
  std::memcpy(&tb, c->h_u64 + 7, 4);
  std::memcpy(&ml, c->h_u64 + 9, 4);
  *max_len = (ml >= 1 && ml <= 64) ? (int)ml : 64;
  if (tb < 6 || tb > 1022 || (tb & 1)) return fail(c, GHF_E_FORMAT, "bad tree_bytes");
  *hdr = (size_t)tb + 2;
  if (stream_bytes < *hdr || (left_bits && stream_bytes == *hdr)) return fail(c, GHF_E_FORMAT, "stream shorter than its header");
  *end_bit = (uint64_t)stream_bytes * 8 - (uint64_t)left_bits;
  return GHF_OK;
}

int ghf_crs_decoded_size(ghf_ctx* c, const uint8_t* d_stream, size_t stream_bytes, int left_bits, const ghf_tree* d_tree,
                         uint64_t* n_out) {
  if (!c || !d_stream || !d_tree || !n_out) return GHF_E_INVAL;
  if (!aligned16(d_stream)) return fail(c, GHF_E_INVAL, "d_stream must be 16-byte aligned");
  GHF_HIP(c, hipSetDevice(c->device));
  size_t hdr;
  uint64_t end_bit;
  int depth = 64;
  int rc = crs_geometry(c, d_tree, stream_bytes, left_bits, &hdr, &end_bit, &depth);
  if (rc) return rc;
  launch_crs_decode_tables(d_tree, c->d_dt, c->d_status, c->stream);
  c->dt_code = nullptr;
  if (end_bit == (uint64_t)hdr * 8) {  // an empty body decodes to nothing
    *n_out = 0;
    return GHF_OK;
  }
  rc = rebuild_index_at(c, d_stream, stream_bytes, hdr, end_bit, 1, (size_t)-1, 0, nullptr, nullptr, false, depth);  // (codes up to 64 bits)
  if (rc) return rc;
  *n_out = c->fidx.n_symbols;
  return GHF_OK;
}

int ghf_crs_sync_piece(ghf_ctx* c, const uint8_t* d_piece, size_t piece_bytes, uint32_t first_bit, uint64_t end_bit,
                       const ghf_tree* d_tree, uint64_t* landing, uint64_t* n_symbols) {
  if (!c || !d_piece || !d_tree || !landing || !n_symbols) return GHF_E_INVAL;
  if (!aligned16(d_piece)) return fail(c, GHF_E_INVAL, "d_piece must be 16-byte aligned");
  if (first_bit >= 512 || end_bit > (uint64_t)piece_bytes * 8 || first_bit > end_bit) return fail(c, GHF_E_INVAL, "ghf_crs_sync_piece: bad first_bit / end_bit");
  GHF_HIP(c, hipSetDevice(c->device));
  launch_crs_decode_tables(d_tree, c->d_dt, c->d_status, c->stream);
  c->dt_code = nullptr;
  c->fidx_stream = nullptr;
  *landing = 0;
  *n_symbols = 0;
  if (end_bit == 0) {  // nothing of this piece is its own
    c->fidx.n_symbols = 0;
    c->fidx.n_segs = 0;
    c->fidx.n_chunks = 0;
    c->fidx_stream = d_piece;
    c->fidx_bytes = piece_bytes;
    return GHF_OK;
  }
  int bad = 0;  // with the tree's tables "end mark" can only mean: bits that are no code
  int depth = 64;
  int rc = crs_max_len(c, d_tree, &depth);
  if (rc) return rc;
  rc = rebuild_index_at(c, d_piece, piece_bytes, 0, end_bit, 2, (size_t)-1, first_bit, landing, &bad, false, depth);
  if (rc) return rc;
  if (bad) return fail(c, GHF_E_CORRUPT, "the .crs body holds bits that are no code");
  *n_symbols = c->fidx.n_symbols;
  return GHF_OK;
}

int ghf_crs_decode(ghf_ctx* c, const uint8_t* d_stream, size_t stream_bytes, int left_bits, const ghf_tree* d_tree,
                   const ghf_index* index, uint8_t* d_out, size_t cap, uint64_t* d_out_bytes) {
  if (!c || !d_stream || !d_tree || !d_out) return GHF_E_INVAL;
  if (!aligned16(d_stream)) return fail(c, GHF_E_INVAL, "d_stream must be 16-byte aligned");
  GHF_HIP(c, hipSetDevice(c->device));
  launch_crs_decode_tables(d_tree, c->d_dt, c->d_status, c->stream);
  c->dt_code = nullptr;
  if (!index) {
    if (c->fidx_stream != d_stream || c->fidx_bytes != stream_bytes) {  // else: ghf_crs_decoded_size / ghf_crs_sync_piece did it
      size_t hdr;
      uint64_t end_bit;
      int depth = 64;
      int rc = crs_geometry(c, d_tree, stream_bytes, left_bits, &hdr, &end_bit, &depth);
      if (rc) return rc;
      if (end_bit == (uint64_t)hdr * 8) {
        if (d_out_bytes) launch_store_u64(d_out_bytes, nullptr, 0, c->stream);
        return GHF_OK;
      }
      rc = rebuild_index_at(c, d_stream, stream_bytes, hdr, end_bit, 1, cap, 0, nullptr, nullptr, false, depth);
      if (rc) return rc;
    }
    c->fidx_stream = nullptr;
    index = &c->fidx;
    if (index->n_symbols == 0) {
      if (d_out_bytes) launch_store_u64(d_out_bytes, nullptr, 0, c->stream);
      return GHF_OK;
    }
  }
  if (!index->d_chunk_bit || !index->d_seg_bit || index->seg_symbols != (uint32_t)kSegSymbols ||
      index->chunk_symbols != (uint32_t)kBlockSymbols || index->n_segs != (index->n_symbols + kSegSymbols - 1) / kSegSymbols ||
      index->n_chunks != (index->n_symbols + kBlockSymbols - 1) / kBlockSymbols)
    return fail(c, GHF_E_INVAL, "ghf_crs_decode: malformed index");
  if (cap < index->n_symbols) return fail(c, GHF_E_CAP, "ghf_crs_decode: output capacity below n_symbols");
  if (index->n_chunks >= kDecMaxGroups) return fail(c, GHF_E_INVAL, "ghf_crs_decode: more than 2^44 symbols in one call");
  DecParams p;
  p.stream = d_stream;
  p.stream_bytes = stream_bytes;
  p.dt = c->d_dt;
  p.chunk_bit = index->d_chunk_bit;
  p.seg_bit = index->d_seg_bit;
  p.n_symbols = index->n_symbols;
  p.n_segs = index->n_segs;
  p.no_end_mark = 1u;  // there is none in this format; the end of every segment but the last is checked against the side-car
  p.out = d_out;
  p.status = c->d_status;
  p.out_bytes = d_out_bytes;
  launch_decode(p, c->stream);
  if (d_out_bytes && index->n_symbols == 0) launch_store_u64(d_out_bytes, nullptr, 0, c->stream);
  GHF_HIP(c, hipGetLastError());
  return GHF_OK;
}

}  // extern "C"
