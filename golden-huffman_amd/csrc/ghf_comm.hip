// golden-huffman_amd/csrc/ghf_comm.hip -- SURVEY 8(e): the two exchanges that make per-GPU shards ONE .crs2 stream, on RCCL
// (xGMI underneath), behind the C ABI of include/ghf.h.  Both messages are tiny (2 KiB, 8 B per rank): latency-bound, so
// they are simply queued on the context's stream between the kernels they separate -- no host synchronisation anywhere.
//
// RCCL is bound at first use with dlopen: a process has ONE copy (PyTorch-ROCm ships its own librccl.so with the SONAME
// of /opt/rocm's), and single-GPU users need none.  The prototypes come from <rccl/rccl.h>.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "ghf_internal.h"

using namespace ghf;

struct ghf_comm {
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0;
};

namespace {

struct Rccl {
  void* so = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string err;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // the copy the process already mapped (RTLD_NOLOAD), else the system one
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
      if (!r.so) r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    for (const char* n : names)
      if (!r.so) r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.so) {
      r.err = std::string("dlopen(librccl): ") + dlerror();
      return;
    }
#define GHF_SYM(name)                                                        \
  r.name = reinterpret_cast<decltype(r.name)>(dlsym(r.so, "nccl" #name));    \
  if (!r.name && r.err.empty()) r.err = "librccl lacks nccl" #name
    GHF_SYM(GetVersion);
    GHF_SYM(GetUniqueId);
    GHF_SYM(CommInitRank);
    GHF_SYM(CommDestroy);
    GHF_SYM(AllReduce);
    GHF_SYM(AllGather);
    GHF_SYM(GetErrorString);
#undef GHF_SYM
  });
  return r.err.empty() ? &r : nullptr;
}

}  // namespace

// from ghf_api.hip
int ghf_api_fail(ghf_ctx* c, int code, const char* what);
hipStream_t ghf_api_stream(ghf_ctx* c);
int ghf_api_device(ghf_ctx* c);
uint64_t* ghf_api_scratch_u64(ghf_ctx* c);  // [8] device scalars
uint64_t* ghf_api_hist(ghf_ctx* c);         // [257] device
uint64_t* ghf_api_totals(ghf_ctx* c, int world);  // [world] device, grown on demand

#define GHF_NCCL(c, call)                                                                         \
  do {                                                                                            \
    ncclResult_t r_ = (call);                                                                     \
    if (r_ != ncclSuccess) return ghf_api_fail((c), GHF_E_HIP, (std::string(#call ": ") + R->GetErrorString(r_)).c_str()); \
  } while (0)

extern "C" {

int ghf_rccl_version(int* version) {
  Rccl* R = rccl();
  if (!R || !version) return GHF_E_HIP;
  return R->GetVersion(version) == ncclSuccess ? GHF_OK : GHF_E_HIP;
}

int ghf_comm_unique_id(uint8_t id[GHF_COMM_ID_BYTES]) {
  static_assert(GHF_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ncclUniqueId travels as GHF_COMM_ID_BYTES bytes");
  Rccl* R = rccl();
  if (!R || !id) return GHF_E_HIP;
  ncclUniqueId u;
  if (R->GetUniqueId(&u) != ncclSuccess) return GHF_E_HIP;
  std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return GHF_OK;
}

int ghf_comm_init_rank(ghf_ctx* c, const uint8_t id[GHF_COMM_ID_BYTES], int world, int rank, ghf_comm** out) {
  if (!c || !id || !out || world < 1 || rank < 0 || rank >= world) return GHF_E_INVAL;
  *out = nullptr;
  Rccl* R = rccl();
  if (!R) return ghf_api_fail(c, GHF_E_HIP, "RCCL is not available in this process");
  if (hipSetDevice(ghf_api_device(c)) != hipSuccess) return ghf_api_fail(c, GHF_E_HIP, "hipSetDevice");
  ghf_comm* m = new (std::nothrow) ghf_comm();
  if (!m) return GHF_E_NOMEM;
  ncclUniqueId u;
  std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  const ncclResult_t r = R->CommInitRank(&m->comm, world, u, rank);
  if (r != ncclSuccess) {
    delete m;
    return ghf_api_fail(c, GHF_E_HIP, (std::string("ncclCommInitRank: ") + R->GetErrorString(r)).c_str());
  }
  m->world = world;
  m->rank = rank;
  *out = m;
  return GHF_OK;
}

int ghf_comm_destroy(ghf_comm* m) {
  if (!m) return GHF_OK;
  Rccl* R = rccl();
  if (R && m->comm) (void)R->CommDestroy(m->comm);
  delete m;
  return GHF_OK;
}

int ghf_comm_world(const ghf_comm* m, int* world, int* rank) {
  if (world) *world = m ? m->world : 1;
  if (rank) *rank = m ? m->rank : 0;
  return GHF_OK;
}

int ghf_comm_allreduce_hist(ghf_ctx* c, ghf_comm* m, uint64_t* d_hist) {
  if (!c || !d_hist) return GHF_E_INVAL;
  if (!m || m->world == 1) return GHF_OK;
  Rccl* R = rccl();
  if (!R) return ghf_api_fail(c, GHF_E_HIP, "RCCL is not available in this process");
  // 256 counts; the end-mark slot [256] == 1 on every rank must not be summed (include/encoder.h:128)
  GHF_NCCL(c, R->AllReduce(d_hist, d_hist, 256, ncclUint64, ncclSum, m->comm, ghf_api_stream(c)));
  return GHF_OK;
}

int ghf_comm_allgather_total(ghf_ctx* c, ghf_comm* m, const uint64_t* d_total, uint64_t* d_totals) {
  if (!c || !d_total || !d_totals) return GHF_E_INVAL;
  if (!m || m->world == 1) {
    launch_store_u64(d_totals, d_total, 0, ghf_api_stream(c));
    return GHF_OK;
  }
  Rccl* R = rccl();
  if (!R) return ghf_api_fail(c, GHF_E_HIP, "RCCL is not available in this process");
  GHF_NCCL(c, R->AllGather(d_total, d_totals, 1, ncclUint64, m->comm, ghf_api_stream(c)));
  return GHF_OK;
}

size_t ghf_shard_bound(size_t n) {
  size_t b = 1040 + 8 * 32 + 4 * n + 8;
  return ((b + 15) & ~(size_t)15) + 32;
}

int ghf_shard_bytes(ghf_ctx* c, const ghf_code* d_code, const uint64_t* d_totals, int world, int rank, size_t* bytes) {
  if (!c || !d_code || !d_totals || !bytes || world < 1 || world > 4096 || rank < 0 || rank >= world) return GHF_E_INVAL;
  if (hipSetDevice(ghf_api_device(c)) != hipSuccess) return ghf_api_fail(c, GHF_E_HIP, "hipSetDevice");
  std::vector<uint64_t> tot((size_t)world);
  int32_t lens[2] = {0, 0};  // min_len, max_len
  uint32_t eof_len = 0;
  hipStream_t s = ghf_api_stream(c);
  hipError_t e = hipMemcpyAsync(tot.data(), d_totals, sizeof(uint64_t) * (size_t)world, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipMemcpyAsync(lens, &d_code->min_len, sizeof lens, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipMemcpyAsync(&eof_len, &d_code->length[GHF_NSYM - 1], sizeof eof_len, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) return ghf_api_fail(c, GHF_E_HIP, "ghf_shard_bytes: copy totals / code lengths to host");
  if (lens[1] < 1 || lens[1] > 32 || eof_len > 32) return ghf_api_fail(c, GHF_E_FORMAT, "ghf_shard_bytes: bad max_len in tables");
  // the arithmetic of K5's emit_begin (ghf_emit.hip): the shard's first bit, its last, the 16-byte units between
  uint64_t start = 8ull * (1040ull + 8ull * (uint64_t)lens[1]);
  for (int h = 0; h < rank; ++h) start += tot[(size_t)h];
  uint64_t end = start + tot[(size_t)rank];
  if (rank == world - 1) end = (end + eof_len + 7) & ~7ull;  // end mark + padding to a byte
  const uint64_t origin = rank > 0 ? ((start >> 7) << 4) : 0ull;  // GHF_EMIT_REBASE: d_out[0] = the unit of the first bit
  *bytes = (size_t)((((end >> 7) + 1) << 4) - origin);
  return GHF_OK;
}

int ghf_encode_sharded(ghf_ctx* c, ghf_comm* m, const uint8_t* d_in, size_t n, uint8_t* d_out, size_t cap, ghf_code* d_code,
                       ghf_index* index, uint64_t* d_start_bit, uint64_t* d_end) {
  if (!c || !d_out || !d_code || (n && !d_in)) return GHF_E_INVAL;
  const int world = m ? m->world : 1, rank = m ? m->rank : 0;
  int rc;
  uint64_t* d_hist = ghf_api_hist(c);
  uint64_t* d_u64 = ghf_api_scratch_u64(c);
  uint64_t* d_totals = ghf_api_totals(c, world);
  if (!d_totals) return GHF_E_NOMEM;
  if ((rc = ghf_histogram(c, d_in, n, d_hist))) return rc;                  // K1, local
  if ((rc = ghf_comm_allreduce_hist(c, m, d_hist))) return rc;               // 2 KiB over xGMI
  if ((rc = ghf_build_code(c, d_hist, d_code))) return rc;                   // K2/K3 redundantly: deterministic, no broadcast
  if ((rc = ghf_encode_plan(c, d_in, n, d_code, d_u64))) return rc;          // K4, local: body bits of this shard
  if ((rc = ghf_comm_allgather_total(c, m, d_u64, d_totals))) return rc;     // 8 B per rank
  uint64_t* start = d_start_bit ? d_start_bit : d_u64 + 6;
  if ((rc = ghf_shard_start_bit(c, d_code, d_totals, world, rank, start))) return rc;
  int flags = 0;
  if (rank == world - 1) flags |= GHF_EMIT_LAST;
  if (rank > 0) flags |= GHF_EMIT_REBASE;
  else flags |= GHF_EMIT_HEADER;
  if (index) index->flags = rank == world - 1 ? 0u : (uint32_t)GHF_INDEX_NO_END_MARK;
  return ghf_encode_emit(c, d_in, n, d_code, rank > 0 ? start : nullptr, flags, d_out, cap, index, d_end);  // K5
}

}  // extern "C"
