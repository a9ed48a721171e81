// golden-huffman_amd/csrc/ghf_device.h -- small device-side helpers shared by the kernel translation units.
#ifndef GHF_DEVICE_H_
#define GHF_DEVICE_H_
#include "ghf_internal.h"

namespace ghf {

__device__ __forceinline__ uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

// LDS traffic between the lanes of ONE wave needs no s_barrier: the LDS executes a wave's
// instructions in order.  This only stops the compiler from moving LDS accesses across the point.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void latch_status(int* st, int code) { atomicCAS(st, 0, code); }

// 16-byte store with the non-temporal hint (`global_store_dwordx4 ... nt`): the line is not kept in the Infinity Cache.
// For data nobody on this GPU reads again soon -- K7's decoded output: otherwise it sits in the cache as dirty lines that
// the NEXT kernel's reads have to evict first (256 MiB pipelined bench: K1 0.093 -> 0.063 ms with this and load_stream).
typedef uint32_t u32x4_stream __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_stream(const void* p) {  // the same hint for data that is read exactly once (K7's compressed span)
  const u32x4_stream x = __builtin_nontemporal_load(reinterpret_cast<const u32x4_stream*>(p));
  return make_uint4(x.x, x.y, x.z, x.w);
}
__device__ __forceinline__ void store_stream(void* p, const uint4& v) {
  const u32x4_stream x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<u32x4_stream*>(p));
}

// inclusive prefix sum over the 64 lanes in six v_add_u32_dpp: row_shr 1/2/4/8 inside the rows of 16, then
// row_bcast:15 and row_bcast:31 carry the row totals across (the classic gfx9 wave64 scan; no LDS traffic,
// unlike __shfl_up, which lowers to ds_bpermute_b32 and costs an LDS round trip per step)
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);  // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1 and 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2 and 3
  return (uint32_t)x;
}

// value of lane 63 as a wave-uniform (SGPR) value: v_readlane_b32, no LDS round trip
__device__ __forceinline__ uint32_t wave_last_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }

// ({hi, lo} >> (s & 31))[31:0] -- v_alignbit_b32
__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t s) { return __builtin_amdgcn_alignbit(hi, lo, s); }

// a5: word w of the .crs2 header.  u32 big-endian: 257, symbol_[0..256], min_len, max_len, (start_pos[i], first_code[i])
// i = 1..max_len  (canonical_huff_encoder.cc:223-237, utils/include/buffer.h:261-268)
__device__ __forceinline__ uint32_t header_word(const ghf_code* code, int w, int max_len) {
  if (w == 0) return GHF_NSYM;
  if (w <= GHF_NSYM) return code->symbol[w - 1];
  if (w == GHF_NSYM + 1) return (uint32_t)code->min_len;
  if (w == GHF_NSYM + 2) return (uint32_t)max_len;
  const int k = w - (GHF_NSYM + 3);
  const int i = 1 + (k >> 1);
  return (k & 1) ? code->first_code[i] : code->start_pos[i];
}

}  // namespace ghf
#endif
