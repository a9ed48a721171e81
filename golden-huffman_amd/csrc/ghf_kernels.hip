// golden-huffman_amd/csrc/ghf_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the
// canonical-Huffman hot path.  No MFMA anywhere: this is HBM/LDS-bound byte and bit work.
//
//   K1  k_histogram        byte histogram + per-chunk histograms   (include/encoder.h:123-150)
//   K2  k_build_code       code lengths on ONE wavefront, libstdc++ heap order emulated exactly
//   K3  (same kernel)      canonical assignment                     (canonical_huff_encoder.cc:69-141,289-345)
//   a5  k_write_header     big-endian .crs2 header                  (canonical_huff_encoder.cc:210-242)
//   K4  k_chunk_bits/k_scan per-chunk bit totals + exclusive scan   (first pass of the two-pass packer)
//   K5  (ghf_emit.hip)
//   K7  k_decode           table-driven block-parallel decode        (canonical_huff_encoder.cc:377-568)
//
// File:line citations are relative to the reference tree (chenghuige/golden-huffman).
#include "ghf_device.h"

namespace ghf {

// ------------------------------------------------------------------------------------------------
// K1: histogram.  bins[256][32] in LDS, replica = lane % 32: in every 32-lane LDS group each lane
// owns its own bank, so ds_add_u32 is conflict-free for ANY byte distribution (16-symbol streams put
// all traffic on 16 bins).  One workgroup walks whole chunks; at each chunk end the 32 replicas are
// summed (running cumulative counters, differences mod 2^32 -> no re-zeroing) and the chunk's 256
// counts are stored: K4 then gets every chunk's bit total without re-reading the input.
// ------------------------------------------------------------------------------------------------
#define GHF_HADD(x, sh) atomicAdd(&lh[((((x) >> (sh)) & 0xFFu) << 5) | rep], 1u)

__device__ __forceinline__ void hist_vec(uint32_t* lh, uint32_t rep, const uint4& v) {
  GHF_HADD(v.x, 0); GHF_HADD(v.x, 8); GHF_HADD(v.x, 16); GHF_HADD(v.x, 24);
  GHF_HADD(v.y, 0); GHF_HADD(v.y, 8); GHF_HADD(v.y, 16); GHF_HADD(v.y, 24);
  GHF_HADD(v.z, 0); GHF_HADD(v.z, 8); GHF_HADD(v.z, 16); GHF_HADD(v.z, 24);
  GHF_HADD(v.w, 0); GHF_HADD(v.w, 8); GHF_HADD(v.w, 16); GHF_HADD(v.w, 24);
}

__global__ __launch_bounds__(kHistThreads) void k_histogram(const uint8_t* __restrict__ in, uint64_t n,
                                                            uint32_t chunk_log2, uint32_t nchunks,
                                                            uint32_t* __restrict__ chunk_hist,
                                                            unsigned long long* __restrict__ hist,
                                                            unsigned long long* __restrict__ acc /* [32][256] + done */) {
  static_assert(kHistRep == 32, "replica index is lane % 32");
  __shared__ uint32_t lh[256 * kHistRep];
  __shared__ bool s_last;
  const uint32_t tid = threadIdx.x;
  const uint32_t rep = tid & 31u;
  for (uint32_t i = tid; i < 256 * kHistRep; i += kHistThreads) lh[i] = 0;
  __syncthreads();
  uint32_t prev = 0;
  unsigned long long total = 0;
  const uint64_t chunk = 1ull << chunk_log2;

  __shared__ uint32_t s_tick;
  // end of a chunk: thread t sums the 32 replicas of bin t (rotated start: 32 lanes on 32 banks); the counters
  // keep running, the chunk's count is the difference to the previous sum (mod 2^32).  Returns the ticket word
  // thread 0 posted before the call (how the workgroup learns its chunk after next without an extra barrier).
  auto finish_chunk = [&](uint32_t c) -> uint32_t {
    __syncthreads();
    const uint32_t posted = s_tick;
    uint32_t s = 0;
#pragma unroll
    for (uint32_t j = 0; j < kHistRep; ++j) s += lh[(tid << 5) | ((j + tid) & 31u)];
    const uint32_t cnt = s - prev;
    prev = s;
    chunk_hist[(uint64_t)c * 256 + tid] = cnt;
    total += cnt;
    __syncthreads();
    return posted;
  };

  // ---- fast phase: full, 16-byte aligned chunks, handed out by ticket counters (16 classes of workgroups, one
  // counter per class on its own 128-byte line: a single counter saturates at ~88 tickets/us).  Dynamic hand-out
  // keeps a late or slow workgroup from becoming the kernel's straggler.  Per thread the vectors of consecutive
  // chunks form ONE stream: four 16-byte loads are always in flight (A/B and C/D alternate, no register copies),
  // also across the chunk boundary -- the next chunk's first vectors are requested while this one is reduced.
  const uint32_t vlog = chunk_log2 - 12;  // vectors per thread per chunk = 2^vlog (256 threads x 16 B = 4 KiB)
  const uint32_t nfullchunks = (uint32_t)(n >> chunk_log2);
  const bool fast = vlog >= 2 && (((uintptr_t)in) & 15u) == 0 && nfullchunks > 0;
  if (fast) {
    const uint32_t ncls = gridDim.x < 16u ? gridDim.x : 16u;
    const uint32_t cls = blockIdx.x % ncls;
    unsigned long long* tick = acc + 32 * 256 + 16 + cls * 16;
    // the first two chunks of a workgroup are fixed, so its first loads go out before any counter has answered;
    // tickets number the chunks behind those 2 * gridDim.x
    auto draw = [&]() -> uint32_t { return 2u * gridDim.x + (uint32_t)atomicAdd(tick, 1ull) * ncls + cls; };  // thread 0 only
    uint32_t cur = blockIdx.x;
    uint32_t nxt = blockIdx.x + gridDim.x;
    const uint32_t V = 1u << vlog;
    auto vptr = [&](uint32_t c, uint32_t j) -> const uint4* {
      if (c >= nfullchunks) c = nfullchunks - 1;  // past the end: redundant, harmless loads
      return reinterpret_cast<const uint4*>(in + ((uint64_t)c << chunk_log2)) + (uint64_t)j * kHistThreads + tid;
    };
    uint4 A = *vptr(cur, 0), B = *vptr(cur, 1);
    while (cur < nfullchunks) {
      uint32_t t_next = 0;
      if (tid == 0) t_next = draw();  // the chunk after next; the atomic returns long before it is needed
      for (uint32_t j = 0; j < V; j += 4) {
        const uint4 C = *vptr(cur, j + 2), D = *vptr(cur, j + 3);
        hist_vec(lh, rep, A);
        hist_vec(lh, rep, B);
        const bool more = j + 4 < V;
        A = *vptr(more ? cur : nxt, more ? j + 4 : 0);
        B = *vptr(more ? cur : nxt, more ? j + 5 : 1);
        hist_vec(lh, rep, C);
        hist_vec(lh, rep, D);
      }
      if (tid == 0) s_tick = t_next;
      const uint32_t posted = finish_chunk(cur);
      cur = nxt;
      nxt = posted;
    }
  }

  // ---- generic phase: small chunks, unaligned input (all chunks, static), or just the ragged last chunk
  const uint32_t g0 = fast ? (blockIdx.x == 0 ? nfullchunks : nchunks) : blockIdx.x;
  const uint32_t gstep = fast ? nchunks : gridDim.x;
  for (uint32_t c = g0; c < nchunks; c += gstep) {
    const uint64_t base = (uint64_t)c << chunk_log2;
    const uint64_t len = (n - base < chunk) ? (n - base) : chunk;
    const uint8_t* p = in + base;
    uint32_t head = (uint32_t)((16u - (uint32_t)((uintptr_t)p & 15u)) & 15u);
    if (head > len) head = (uint32_t)len;
    if (tid < head) atomicAdd(&lh[((uint32_t)p[tid] << 5) | rep], 1u);
    const uint4* pv = reinterpret_cast<const uint4*>(p + head);
    const uint64_t nvec = (len - head) >> 4;
    for (uint64_t i = tid; i < nvec; i += kHistThreads) {
      const uint4 v0 = pv[i];
      hist_vec(lh, rep, v0);
    }
    const uint64_t tail0 = head + (nvec << 4);
    if (tail0 + tid < len) atomicAdd(&lh[((uint32_t)p[tail0 + tid] << 5) | rep], 1u);
    (void)finish_chunk(c);
  }
  // Global totals.  1024 workgroups adding into the same 256 words would serialise at the memory side (one word
  // takes ~12 ns per atomic), so each workgroup adds into one of 32 replicas; the LAST workgroup to finish sums
  // the replicas into the caller's histogram and leaves them zeroed for the next launch (no memset, no extra
  // kernel).  Hand-off: every wave drains its atomics, workgroup barrier, one lane takes a ticket; the last
  // workgroup acquires and reads the replicas with agent-scope (L1-bypassing) loads.
  unsigned long long* rep_base = acc + (uint64_t)(blockIdx.x & 31u) * 256;
  if (total) atomicAdd(&rep_base[tid], total);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    // no release fence: the only data handed over are the atomic adds above, which execute at the memory side
    // and were acknowledged (vmcnt drained) before the barrier; the plain chunk_hist stores are for later kernels
    const unsigned long long t = atomicAdd(&acc[32 * 256], 1ull);
    s_last = (t + 1 == gridDim.x);
    if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (s_last) {
    unsigned long long sum = 0;
#pragma unroll 8
    for (int r = 0; r < 32; ++r) {
      sum += __hip_atomic_load(&acc[r * 256 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&acc[r * 256 + tid], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    hist[tid] = sum;
    if (tid < 16) __hip_atomic_store(&acc[32 * 256 + 16 + tid * 16], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) {
      hist[256] = 1;  // include/encoder.h:128 end-of-stream mark counts once
      __hip_atomic_store(&acc[32 * 256], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

void launch_histogram(const uint8_t* d_in, uint64_t n, uint32_t chunk_log2, uint32_t nchunks, uint32_t* d_chunk_hist,
                      uint64_t* d_hist, uint64_t* d_acc, hipStream_t s) {
  // one resident round of workgroups, 4 per CU: measured faster than the 5 the LDS would allow (2 GiB stream:
  // 5.55 TB/s at 1024 workgroups vs 4.95 TB/s at 1280 -- the fifth workgroup only adds L2/LDS pressure)
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    if (ncu < 1) ncu = 256;
  }
  uint32_t grid = (uint32_t)(4 * ncu);
  if (grid > nchunks) grid = nchunks;
  if (grid == 0) grid = 1;
  hipLaunchKernelGGL(k_histogram, dim3(grid), dim3(kHistThreads), 0, s, d_in, n, chunk_log2, nchunks, d_chunk_hist,
                     reinterpret_cast<unsigned long long*>(d_hist), reinterpret_cast<unsigned long long*>(d_acc));
}

// ------------------------------------------------------------------------------------------------
// K2 + K3: code lengths and canonical assignment on one wavefront.
//
// The reference keeps symbol INDICES in a std::priority_queue ordered by the live frequency table
// (include/canonical_huff_encoder.h:58-70).  Which of several equal-weight nodes is popped first is
// decided by libstdc++'s heap layout, and that decides the code lengths, so the heap is emulated
// step for step (lane 0: __push_heap / __adjust_heap as in <bits/stl_heap.h>).  The "+1 for every
// member of both chains" walks of canonical_huff_encoder.cc:316-329 become a recorded merge tree whose
// leaf depths all 64 lanes read off in parallel afterwards.
// ------------------------------------------------------------------------------------------------
// Heap entry = (frequency << 9) | symbol index in ONE 64-bit word, so a sift level moves one word and a
// parent's two children (and its four grandchildren) are one (two) aligned 16-byte LDS reads.
// The reference's comparator looks at the frequency only -- equal frequencies must compare EQUAL, the
// index must not break ties: comp(a, b) = freq[a] > freq[b]  <=>  ea > (eb | 511).
// Heap position p lives in slot p + 1 so that the child pair (2p+1, 2p+2) sits on a 16-byte boundary.
struct HeapLds {
  alignas(16) unsigned long long slot[528];
  uint16_t parent[GHF_NSYM + 256 + 7];  // Huffman tree: node -> parent node (0 = none); leaves 0..256, merges 257..
  uint16_t cur[GHF_NSYM + 3];           // symbol index kept in the heap -> the tree node it currently stands for
};

typedef unsigned long long u64t;
struct alignas(16) U64x2 { u64t x, y; };

__device__ __forceinline__ bool heap_gt(u64t a, u64t b) { return a > (b | 511ull); }  // freq(a) > freq(b)

__device__ __forceinline__ void heap_sift_up(HeapLds& h, int hole, u64t e) {
  // libstdc++ __push_heap(first, hole, top = 0, value, comp)
  while (hole > 0) {
    const int parent = (hole - 1) >> 1;
    const u64t pe = h.slot[parent + 1];
    if (!heap_gt(pe, e)) break;
    h.slot[hole + 1] = pe;
    hole = parent;
  }
  h.slot[hole + 1] = e;
}

// std::pop_heap + pop_back: a[0] leaves, then __adjust_heap(first, 0, len = n-1, value = old back): the hole
// walks to the bottom always taking the child for which comp(right, left) is false -> right, else left (two
// levels per LDS round trip: the grandchildren are fetched together with the children), the lone left
// child of an even-length heap is handled, then the displaced value is pushed up from the hole.
__device__ __forceinline__ u64t heap_pop(HeapLds& h, int& n) {
  const u64t top = h.slot[1];
  const int len = n - 1;
  n = len;
  if (len < 1) return top;
  const u64t value = h.slot[len + 1];
  int hole = 0;
  const int lim = (len - 1) >> 1;
  while (hole < lim) {
    const U64x2 c = *reinterpret_cast<const U64x2*>(&h.slot[2 * hole + 2]);   // positions 2h+1, 2h+2
    const U64x2 g0 = *reinterpret_cast<const U64x2*>(&h.slot[4 * hole + 4]);  // positions 4h+3, 4h+4
    const U64x2 g1 = *reinterpret_cast<const U64x2*>(&h.slot[4 * hole + 6]);  // positions 4h+5, 4h+6
    const bool left = heap_gt(c.y, c.x);
    const int child = 2 * hole + (left ? 1 : 2);
    h.slot[hole + 1] = left ? c.x : c.y;
    hole = child;
    if (hole < lim) {
      const U64x2 g = left ? g0 : g1;
      const bool left2 = heap_gt(g.y, g.x);
      const int child2 = 2 * hole + (left2 ? 1 : 2);
      h.slot[hole + 1] = left2 ? g.x : g.y;
      hole = child2;
    }
  }
  if ((len & 1) == 0 && hole == ((len - 2) >> 1)) {
    const int child = 2 * (hole + 1);
    h.slot[hole + 1] = h.slot[child];  // position child - 1
    hole = child - 1;
  }
  heap_sift_up(h, hole, value);
  return top;
}

// ---- the same two heap operations, executed by the WHOLE wave in a constant number of steps ----
// A sift only ever touches one root-to-leaf path, and libstdc++'s __adjust_heap picks that path from
// sibling comparisons alone (the value being re-inserted plays no part until the final __push_heap).  So:
//   1. every lane compares the two children of "its" internal nodes -> two ballots = a 128-bit map of
//      preferred children for the whole heap;
//   2. the scalar unit follows the map from the root to a leaf (bit tests, no memory);
//   3. lane j loads the entry at depth j of that path; one ballot against the re-inserted value tells where
//      __push_heap stops; lanes shift their entries up by one (DPP) and one lane drops the value in.
// Identical result to the sequential code above, ~2.5x fewer cycles per pop; heap_pop/heap_sift_up stay as
// the executable specification (GHF_K2_SEQUENTIAL builds use them).
__device__ __forceinline__ u64t lane_above(u64t x) {  // value held by lane + 1 (same row of 16 lanes)
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, 0x101, 0xF, 0xF, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x >> 32), 0x101, 0xF, 0xF, false);
  return ((u64t)hi << 32) | lo;
}

__device__ __forceinline__ u64t wave_uniform(u64t x) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32));
  return ((u64t)hi << 32) | lo;
}

__device__ __forceinline__ u64t wave_heap_pop(HeapLds& h, int& n, int lane) {
  const int len = n - 1;
  n = len;
  const u64t top = wave_uniform(h.slot[1]);
  if (len < 1) return top;
  const u64t value = wave_uniform(h.slot[len + 1]);
  const int lim = (len - 1) >> 1;
  // 1. preferred child of every internal node (nodes lane and lane + 64): 1 = left
  const U64x2 c0 = *reinterpret_cast<const U64x2*>(&h.slot[2 * lane + 2]);
  const U64x2 c1 = *reinterpret_cast<const U64x2*>(&h.slot[2 * (lane + 64) + 2]);
  const unsigned long long m0 = __ballot(heap_gt(c0.y, c0.x));
  const unsigned long long m1 = __ballot(heap_gt(c1.y, c1.x));
  // 2. walk (scalar).  With D = depth of the last position, every node above depth D-1 has two children, so the
  //    first D-1 steps need no bounds test, and all of them but a possible 7th stay below node 63 (map m0 only).
  const int D = 31 - __clz(len);
  const int steps = D - 1;
  int hole = 0, k = 0;
#pragma unroll
  for (int d = 0; d < 6; ++d) {
    if (d < steps) {
      hole = 2 * hole + 2 - (int)((m0 >> hole) & 1ull);
      k = d + 1;
    }
  }
  if (steps > 6) {  // len >= 256: one step from depth 6 (nodes 63..126)
    const unsigned long long bit = (hole < 64 ? (m0 >> hole) : (m1 >> (hole - 64))) & 1ull;
    hole = 2 * hole + 2 - (int)bit;
    ++k;
  }
  if (hole < lim) {  // depth D-1 -> D where that node has both children
    const unsigned long long bit = (hole < 64 ? (m0 >> hole) : (m1 >> (hole - 64))) & 1ull;
    hole = 2 * hole + 2 - (int)bit;
    ++k;
  }
  if ((len & 1) == 0 && hole == ((len - 2) >> 1)) {  // lone left child
    hole = 2 * hole + 1;
    ++k;
  }
  // 3. lane j <-> depth j of the path
  const bool on = lane <= k;
  const int pj = on ? (((hole + 1) >> (k - lane)) - 1) : 0;
  const u64t e = on ? h.slot[pj + 1] : 0ull;
  const bool stop = lane >= 1 && on && !heap_gt(e, value);  // __push_heap stops below this entry
  const unsigned long long sm = __ballot(stop);
  const int m = sm ? 63 - __clzll((long long)sm) : 0;
  const u64t up = lane_above(e);
  if (lane <= m) h.slot[pj + 1] = (lane < m) ? up : value;
  return top;
}

// priority_queue::push(e) onto a heap of n entries: __push_heap from position n
__device__ __forceinline__ void wave_heap_push(HeapLds& h, int n, u64t e, int lane) {
  const int depth = 31 - __clz(n + 1);  // number of ancestors of position n
  const bool on = lane >= 1 && lane <= depth;
  const int aj = ((n + 1) >> lane) - 1;  // lane 0: n itself
  const u64t pe = on ? h.slot[aj + 1] : 0ull;
  const bool stop = on && !heap_gt(pe, e);
  const unsigned long long sm = __ballot(stop);
  const int t = sm ? (__ffsll((long long)sm) - 1) - 1 : depth;  // entries of lanes 1..t move down one level
  const u64t up = lane_above(pe);
  if (lane <= t) h.slot[aj + 1] = (lane < t) ? up : e;
}

struct CodeLds {  // the small per-length tables of K3; the per-symbol arrays go straight to global memory
  uint32_t num[40];
  uint32_t first_code[64];
  uint32_t start_pos[64];
  int32_t min_len, max_len;
};

// SURVEY 8(f) N4, opt-in (GHF_CODE_LIMIT): where the reference cannot go (a code longer than 32 bits,
// include/canonical_huff_encoder.h:43-44) the lengths are replaced by the optimal 32-bit-limited ones
// (package-merge; definition and tie rules: oracle/huff_oracle.c orc_limit_lengths).  Rare and small (<= 257 leaves,
// 32 levels): ranking is done by all lanes, the merges by lane 0.
struct LimitLds {
  unsigned long long w[2][2 * GHF_NSYM];
  uint8_t is_leaf[33][2 * GHF_NSYM];
  uint16_t order[GHF_NSYM + 3];
  uint16_t len_of[34], taken[34];
  uint32_t newlen[GHF_NSYM + 3];
  long long freq[GHF_NSYM + 3];
  int n;
};

__device__ void limit_lengths_32(LimitLds& Q, const uint32_t (&len)[5], int lane) {
  constexpr int kLimit = 32;
  const long long* freq = Q.freq;
  // order: present symbols by (frequency ascending, index ascending)
  if (lane == 0) Q.n = 0;
  __syncthreads();
#pragma unroll 1
  for (int j = 0; j < 5; ++j) {
    const int s = lane + 64 * j;
    if (s >= GHF_NSYM || len[j] == 0) continue;
    const long long f = freq[s];
    int rank = 0;
    for (int t = 0; t < GHF_NSYM; ++t) {
      const long long g = freq[t];
      rank += (g != 0) && (g < f || (g == f && t < s));
    }
    Q.order[rank] = (uint16_t)s;
    atomicAdd(&Q.n, 1);
  }
  __syncthreads();
  if (lane == 0) {
    const int n = Q.n;
    int prev_n = 0, cur = 0;
    for (int d = kLimit; d >= 1; --d) {
      const unsigned long long* pw = Q.w[cur ^ 1];
      unsigned long long* cw = Q.w[cur];
      const int npk = prev_n / 2;
      int li = 0, pi = 0, k = 0;
      while (li < n || pi < npk) {
        const unsigned long long lw = li < n ? (unsigned long long)freq[Q.order[li]] : ~0ull;
        const unsigned long long pk = pi < npk ? pw[2 * pi] + pw[2 * pi + 1] : ~0ull;
        if (li < n && (pi >= npk || lw <= pk)) {
          cw[k] = lw;
          Q.is_leaf[d][k] = 1;
          ++li;
        } else {
          cw[k] = pk;
          Q.is_leaf[d][k] = 0;
          ++pi;
        }
        ++k;
      }
      Q.len_of[d] = (uint16_t)k;
      prev_n = k;
      cur ^= 1;
    }
    int need = 2 * n - 2;
    for (int d = 1; d <= kLimit; ++d) {
      if (need > (int)Q.len_of[d]) need = Q.len_of[d];
      int leaves = 0;
      for (int k = 0; k < need; ++k) leaves += Q.is_leaf[d][k];
      Q.taken[d] = (uint16_t)leaves;
      need = 2 * (need - leaves);
    }
  }
  __syncthreads();
  for (int i = lane; i < GHF_NSYM; i += 64) Q.newlen[i] = 0;
  __syncthreads();
  for (int i = lane; i < Q.n; i += 64) {
    uint32_t l = 0;
    for (int d = 1; d <= kLimit; ++d) l += i < (int)Q.taken[d];
    Q.newlen[Q.order[i]] = l;
  }
  __syncthreads();
}

// LIMIT = the GHF_CODE_LIMIT instantiation: it alone carries LimitLds (26 KiB).  The default one stays small enough
// in LDS to be scheduled next to the streaming kernels of a pipelined caller instead of waiting for a CU to drain.
struct NoLimitLds {};
template <bool LIMIT>
__global__ __launch_bounds__(64) void k_build_code(const unsigned long long* __restrict__ hist, ghf_code* __restrict__ out,
                                                   int* __restrict__ status) {
  __shared__ HeapLds heap;
  __shared__ typename std::conditional<LIMIT, LimitLds, NoLimitLds>::type Q;
  // one latency-bound wave among thousands of streaming ones (the other kernels of a pipelined caller share its CU):
  // let the instruction arbiter of its SIMD prefer it
  __builtin_amdgcn_s_setprio(3);
  __shared__ CodeLds cl;
  __shared__ int s_ndata;
  const int lane = threadIdx.x;
  // the 257 counts live in registers (lane l holds symbols l, l + 64, ..): LDS is kept under 7 KiB so that this wave
  // fits on a CU next to K7's 153 KiB (or K5's, or K1's) instead of waiting for one of their workgroups to retire
  long long fr[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int s = lane + 64 * j;
    fr[j] = s < GHF_NSYM ? (long long)hist[s] : 0ll;
    if constexpr (LIMIT) {
      if (s < GHF_NSYM) Q.freq[s] = fr[j];
    }
  }
  for (int s = lane; s < GHF_NSYM; s += 64) heap.cur[s] = (uint16_t)s;
  for (int i = lane; i < GHF_NSYM + 256 + 7; i += 64) heap.parent[i] = 0;
  if (lane < 40) cl.num[lane] = 0;
  cl.first_code[lane] = 0;
  cl.start_pos[lane] = 0;
  __syncthreads();

  // ---- K2: get_encoding_length, canonical_huff_encoder.cc:289-345.  The ORDER of heap operations is strictly
  // sequential (which of several equal-weight nodes pops first is decided by the heap layout), but each single
  // operation is done by all 64 lanes at once; the merges are recorded as a tree and the depths read off later.
  {
    int n = 0, ndata = 0;
    for (int s = 0; s < GHF_NSYM; ++s) {  // .cc:301-306: ascending index, zero counts skipped
      const u64t f = (u64t)__shfl(fr[s >> 6], s & 63, 64);  // wave-uniform: readlane
      if (f) {
        wave_heap_push(heap, n, (f << 9) | (u64t)s, lane);  // priority_queue::push
        ++n;
        if (s < 256) ++ndata;
      }
    }
    if (lane == 0) s_ndata = ndata;
    const int times = n - 1;  // .cc:309
    for (int t = 0; t < times; ++t) {
      const u64t e1 = wave_heap_pop(heap, n, lane);  // .cc:311-314
      const u64t e2 = wave_heap_pop(heap, n, lane);
      const int s1 = (int)(e1 & 511u), s2 = (int)(e2 & 511u);
      const int node = GHF_NSYM + t;
      if (lane == 0) {
        heap.parent[heap.cur[s1]] = (uint16_t)node;  // .cc:316-329: both groups one level deeper ...
        heap.parent[heap.cur[s2]] = (uint16_t)node;
        heap.cur[s2] = (uint16_t)node;               // ... and merged under the second popped index
      }
      const u64t f = (e1 >> 9) + (e2 >> 9);                  // .cc:331
      wave_heap_push(heap, n, (f << 9) | (u64t)s2, lane);    // .cc:333
      ++n;
    }
  }
  __syncthreads();
  if (s_ndata == 0) {  // empty input: undefined in the reference
    if (lane == 0) latch_status(status, GHF_E_EMPTY);
    return;
  }
  // code length = depth of the leaf; symbols owned by this lane: s_j = lane + 64 j (s = 256 is lane 0, j = 4)
  uint32_t len[5], node[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    len[j] = 0;
    node[j] = (uint32_t)(lane + 64 * j);
    if (node[j] >= GHF_NSYM) node[j] = GHF_NSYM + 256 + 1;  // parent == 0 there
  }
  for (int step = 0; step < 256; ++step) {
    bool any = false;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const uint32_t p = heap.parent[node[j]];
      if (p) {
        node[j] = p;
        len[j] += 1;
        any = true;
      }
    }
    if (!__ballot(any)) break;
  }
  uint32_t mx = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    if (lane + 64 * j >= GHF_NSYM) len[j] = 0;  // lanes past symbol 256 own nothing
    mx = len[j] > mx ? len[j] : mx;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t o = __shfl_xor(mx, d, 64);
    mx = o > mx ? o : mx;
  }
  int max_len = (int)mx;  // .cc:343
  if (max_len > 32) {     // include/canonical_huff_encoder.h:43-44: the reference cannot write such codes
    if constexpr (!LIMIT) {
      if (lane == 0) latch_status(status, GHF_E_CODELEN);
      return;
    } else {
      limit_lengths_32(Q, len, lane);
      mx = 0;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int s = lane + 64 * j;
        len[j] = s < GHF_NSYM ? Q.newlen[s] : 0u;
        mx = len[j] > mx ? len[j] : mx;
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = __shfl_xor(mx, d, 64);
        mx = o > mx ? o : mx;
      }
      max_len = (int)mx;
    }
  }

  // ---- K3: do_gen_encode, canonical_huff_encoder.cc:69-141 ----
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int s = lane + 64 * j;
    if (s < GHF_NSYM) {
      out->length[s] = len[j];
      out->codeword[s] = 0;
      out->symbol[s] = 0xFFFFFFFFu;  // .cc:88
      if (len[j]) atomicAdd(&cl.num[len[j]], 1u);  // .cc:85-87
    }
  }
  __syncthreads();  // (also orders the symbol[] defaults above before the slots written below)
  const uint32_t num = (lane >= 1 && lane <= max_len) ? cl.num[lane] : 0u;
  const unsigned long long nzmask = __ballot(num != 0);
  const int min_len = __ffsll((long long)nzmask) - 1;                 // .cc:93-98
  const uint32_t spos = wave_incl_scan_u32(num) - num;          // .cc:104-105 start_pos[i] = sum num[1..i-1]
  if (lane >= 1 && lane <= max_len) cl.start_pos[lane] = spos;
  if (lane == 0) {                                                     // .cc:109-121
    uint32_t fc = 0;
    cl.first_code[max_len] = 0;
    for (int i = max_len - 1; i >= 1; --i) {
      fc = (fc + cl.num[i + 1]) >> 1;
      cl.first_code[i] = fc;
    }
    for (int i = 1; i < min_len; ++i) cl.first_code[i] = 1024;
    out->min_len = min_len;
    out->max_len = max_len;
  }
  __syncthreads();
  out->first_code[lane] = cl.first_code[lane];  // 64 entries each, zero beyond max_len
  out->start_pos[lane] = cl.start_pos[lane];
  // .cc:127-133: within a length, codes and symbol_[] slots go to symbols in ascending index order.
  // rank = (#same-length symbols in earlier 64-symbol rows) + (#same-length lanes below me in my row)
  uint32_t seen = 0;  // lane L holds how many symbols of length L were ranked so far
  for (int j = 0; j < 5; ++j) {
    for (int L = min_len; L <= max_len; ++L) {
      const bool m = (len[j] == (uint32_t)L);
      const unsigned long long mask = __ballot(m);
      if (mask == 0) continue;
      const uint32_t before = __shfl(seen, L, 64);
      if (m) {
        const uint32_t r = before + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        const int s = lane + 64 * j;
        out->codeword[s] = cl.first_code[L] + r;
        out->symbol[cl.start_pos[L] + r] = (uint32_t)s;
      }
      if (lane == L) seen += (uint32_t)__popcll(mask);
    }
  }
}

void launch_build_code(const uint64_t* d_hist, ghf_code* d_code, int* d_status, uint32_t flags, hipStream_t s) {
  if (flags & GHF_CODE_LIMIT)
    hipLaunchKernelGGL(k_build_code<true>, dim3(1), dim3(64), 0, s, reinterpret_cast<const unsigned long long*>(d_hist), d_code, d_status);
  else
    hipLaunchKernelGGL(k_build_code<false>, dim3(1), dim3(64), 0, s, reinterpret_cast<const unsigned long long*>(d_hist), d_code, d_status);
}

// ------------------------------------------------------------------------------------------------
// SURVEY 8(f) N3, the .crs format: EncodeHuffTree::build_tree + gen_encode + serialize_tree
// (include/huff_tree.h:228-235, include/huff_tree.cc:138-187) on one wavefront.  The queue is the same
// libstdc++ heap on the weight alone as K2's, so the same wave-parallel push/pop applies; what differs is
// what is kept: the tree itself (left = first popped, right = second popped), because the codes are the
// root-to-leaf paths and the file header is the tree in preorder.
// ------------------------------------------------------------------------------------------------
struct TreeLds {
  uint16_t tl[256], tr[256];             // children of merge t (tree node ids of HeapLds: 0..255 leaves, 257 + t merges)
  uint16_t size[GHF_NSYM + 256 + 7];     // nodes in the subtree
  uint8_t is_right[GHF_NSYM + 256 + 7];  // the node is its parent's right child
  uint32_t max_len, min_len;
  int n;
};

__global__ __launch_bounds__(64) void k_crs_build_code(const unsigned long long* __restrict__ hist, ghf_tree* __restrict__ out_tree,
                                                       ghf_code* __restrict__ out_code, unsigned long long* __restrict__ start_bit,
                                                       int* __restrict__ status) {
  __shared__ HeapLds heap;
  __shared__ TreeLds T;
  __shared__ ghf_tree tree;
  __builtin_amdgcn_s_setprio(3);  // see k_build_code
  const int lane = threadIdx.x;
  for (int s = lane; s < GHF_NSYM; s += 64) heap.cur[s] = (uint16_t)s;
  for (int i = lane; i < GHF_NSYM + 256 + 7; i += 64) {
    heap.parent[i] = 0;
    T.size[i] = 1;
    T.is_right[i] = 0;
  }
  for (int i = lane; i < (int)(sizeof(ghf_tree) / 4); i += 64) reinterpret_cast<uint32_t*>(&tree)[i] = 0;
  if (lane == 0) {
    T.max_len = 0;
    T.min_len = 0xFFFFFFFFu;
  }
  __syncthreads();
  int n = 0;
  for (int s = 0; s < 256; ++s) {  // huff_tree.h:228-235: keys ascending, zero counts skipped, no end mark
    const u64t f = wave_uniform((u64t)hist[s]);
    if (f) {
      wave_heap_push(heap, n, (f << 9) | (u64t)s, lane);
      ++n;
    }
  }
  const int nleaves = n;
  if (nleaves < 2) {
    if (lane == 0) latch_status(status, nleaves == 0 ? GHF_E_EMPTY : GHF_E_SINGLE);
    return;
  }
  const int times = n - 1;  // huff_tree.cc:141
  for (int t = 0; t < times; ++t) {
    const u64t e1 = wave_heap_pop(heap, n, lane);  // lchild, huff_tree.cc:143-144
    const u64t e2 = wave_heap_pop(heap, n, lane);  // rchild, :145-146
    const int s1 = (int)(e1 & 511u), s2 = (int)(e2 & 511u);
    const int node = GHF_NSYM + t;
    if (lane == 0) {
      const int a = heap.cur[s1], b = heap.cur[s2];
      heap.parent[a] = (uint16_t)node;
      heap.parent[b] = (uint16_t)node;
      T.tl[t] = (uint16_t)a;
      T.tr[t] = (uint16_t)b;
      T.is_right[b] = 1;
      T.size[node] = (uint16_t)(1 + T.size[a] + T.size[b]);
      heap.cur[s2] = (uint16_t)node;  // the parent travels through the heap under the second child's slot
    }
    wave_heap_push(heap, n, (((e1 >> 9) + (e2 >> 9)) << 9) | (u64t)s2, lane);  // :147-148, weight = sum (huff_tree.h:62-66)
    ++n;
  }
  __syncthreads();
  // every node walks up to the root: depth = code length, the turns taken = the code (root bit first), and
  // 1 + (left sibling's subtree, when coming from the right) per step = its preorder position
  for (int v = lane; v < GHF_NSYM + times; v += 64) {
    const bool leaf = v < 256;
    if (v == 256 || (leaf && heap.parent[v] == 0)) continue;  // the end-mark slot / absent symbols
    uint32_t len = 0, pre = 0;
    unsigned long long code = 0;
    int x = v;
    for (int p; (p = heap.parent[x]) != 0; x = p) {
      const uint32_t r = T.is_right[x];
      if (len < 64) code |= (unsigned long long)r << len;
      ++len;
      pre += 1u + (r ? (uint32_t)T.size[T.tl[p - GHF_NSYM]] : 0u);
    }
    if (leaf) {
      tree.header[2 * pre] = 0;            // huff_tree.cc:178-181
      tree.header[2 * pre + 1] = (uint8_t)v;
      atomicMax(&T.max_len, len);
      atomicMin(&T.min_len, len);
      if (len <= 32) {
        out_code->length[v] = len;
        out_code->codeword[v] = (uint32_t)code;
      }
    } else {
      tree.header[2 * pre] = 255;          // huff_tree.cc:183-184
      tree.header[2 * pre + 1] = 255;
      const int t = v - GHF_NSYM;
      const int a = T.tl[t], b = T.tr[t];
      tree.left[t] = (uint16_t)(a < GHF_NSYM ? a : 256 + (a - GHF_NSYM));
      tree.right[t] = (uint16_t)(b < GHF_NSYM ? b : 256 + (b - GHF_NSYM));
    }
  }
  // symbols that do not occur, the end-mark slot and the canonical-only tables of ghf_code
  for (int s = lane; s < GHF_NSYM; s += 64) {
    if (s == 256 || heap.parent[s] == 0) {
      out_code->length[s] = 0;
      out_code->codeword[s] = 0;
    }
    out_code->symbol[s] = 0xFFFFFFFFu;
  }
  for (int i = lane; i < 64; i += 64) {
    out_code->first_code[i] = 0;
    out_code->start_pos[i] = 0;
  }
  __syncthreads();
  if (T.max_len > 32u) {  // K5 packs codes of at most 32 bits
    if (lane == 0) latch_status(status, GHF_E_CODELEN);
    return;
  }
  if (lane == 0) {
    tree.root = (uint32_t)(256 + times - 1);
    tree.n_leaves = (uint32_t)nleaves;
    tree.max_len = T.max_len;
    tree.tree_bytes = 2u * (2u * (uint32_t)nleaves - 1u);
    out_code->min_len = (int32_t)T.min_len;
    out_code->max_len = (int32_t)T.max_len;
    if (start_bit) *start_bit = 8ull * ((unsigned long long)tree.tree_bytes + 2ull);  // normal_huff_encoder.h:163-164
  }
  __syncthreads();
  for (int i = lane; i < (int)(sizeof(ghf_tree) / 4); i += 64)
    reinterpret_cast<uint32_t*>(out_tree)[i] = reinterpret_cast<const uint32_t*>(&tree)[i];
}

void launch_crs_build_code(const uint64_t* d_hist, ghf_tree* d_tree, ghf_code* d_code, uint64_t* d_start_bit, int* d_status,
                           hipStream_t s) {
  hipLaunchKernelGGL(k_crs_build_code, dim3(1), dim3(64), 0, s, reinterpret_cast<const unsigned long long*>(d_hist), d_tree,
                     d_code, reinterpret_cast<unsigned long long*>(d_start_bit), d_status);
}

// The framing around the body (normal_huff_encoder.h:136-138,159-186): the tree, then {left_bits, last byte}.  K5 has
// written the code bits from byte tree_bytes + 2 on, zero-filled to the end of their last byte; that byte is copied
// into the prefix and no longer counted (the reference stores whole bytes only and seeks back for the rest).
__global__ __launch_bounds__(256) void k_crs_finish(const ghf_tree* __restrict__ tree, const unsigned long long* __restrict__ total_bits,
                                                    uint8_t* __restrict__ out, unsigned long long* __restrict__ out_bytes,
                                                    const int* __restrict__ status) {
  if (*status != 0) return;
  const uint32_t tb = tree->tree_bytes;
  for (uint32_t i = threadIdx.x; i < tb; i += 256) out[i] = tree->header[i];
  if (threadIdx.x == 0) {
    const unsigned long long t = *total_bits;
    const unsigned long long whole = t >> 3;
    const uint32_t left = (uint32_t)((8u - (uint32_t)(t & 7u)) & 7u);
    out[tb] = (uint8_t)left;
    out[tb + 1] = left ? out[tb + 2 + whole] : (uint8_t)0;
    if (out_bytes) *out_bytes = (unsigned long long)tb + 2ull + whole;
  }
}

void launch_crs_finish(const ghf_tree* d_tree, const uint64_t* d_total_bits, uint8_t* d_out, uint64_t* d_out_bytes, int* d_status,
                       hipStream_t s) {
  hipLaunchKernelGGL(k_crs_finish, dim3(1), dim3(256), 0, s, d_tree, reinterpret_cast<const unsigned long long*>(d_total_bits), d_out,
                     reinterpret_cast<unsigned long long*>(d_out_bytes), d_status);
}

// ------------------------------------------------------------------------------------------------
// a5: header.  u32 big-endian: 257, symbol_[0..256], min_len, max_len, (start_pos[i], first_code[i]) i=1..max_len
// (canonical_huff_encoder.cc:223-237, utils/include/buffer.h:261-268)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_write_header(const ghf_code* __restrict__ code, uint8_t* __restrict__ out,
                                                      uint64_t cap, int* __restrict__ status) {
  const int max_len = code->max_len;
  if (max_len < 1 || max_len > 32) return;  // build_code latched the reason
  const int nwords = 1 + GHF_NSYM + 2 + 2 * max_len;
  if ((uint64_t)nwords * 4 > cap) {
    if (threadIdx.x == 0) latch_status(status, GHF_E_CAP);
    return;
  }
  uint32_t* o = reinterpret_cast<uint32_t*>(out);
  for (int w = threadIdx.x; w < nwords; w += blockDim.x) o[w] = bswap32(header_word(code, w, max_len));
}

void launch_write_header(const ghf_code* d_code, uint8_t* d_out, uint64_t cap, int* d_status, hipStream_t s) {
  hipLaunchKernelGGL(k_write_header, dim3(1), dim3(256), 0, s, d_code, d_out, cap, d_status);
}

// ------------------------------------------------------------------------------------------------
// K4: first pass of the two-pass packer.  bits(chunk) = sum_s hist_chunk[s] * length[s]; no input
// re-read when K1 left the per-chunk histograms (k_chunk_bits), else straight from the bytes
// (k_chunk_bits_direct).  Then one workgroup scans the <= a few thousand chunk totals.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_chunk_bits(const uint32_t* __restrict__ chunk_hist, uint32_t nchunks,
                                                    const ghf_code* __restrict__ code, uint64_t* __restrict__ bits) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint4 L = reinterpret_cast<const uint4*>(code->length)[lane];  // lengths of symbols 4*lane .. 4*lane+3
  for (uint32_t c = wave; c < nchunks; c += nwaves) {
    const uint4 h = reinterpret_cast<const uint4*>(chunk_hist + (uint64_t)c * 256)[lane];
    unsigned long long b = (unsigned long long)h.x * L.x + (unsigned long long)h.y * L.y +
                           (unsigned long long)h.z * L.z + (unsigned long long)h.w * L.w;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) b += __shfl_xor(b, d, 64);
    if (lane == 0) bits[c] = b;
  }
}

__global__ __launch_bounds__(256) void k_chunk_bits_direct(const uint8_t* __restrict__ in, uint64_t n, uint32_t chunk_log2,
                                                           uint32_t nchunks, const ghf_code* __restrict__ code,
                                                           uint64_t* __restrict__ bits) {
  __shared__ uint32_t ll[256];
  ll[threadIdx.x] = code->length[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint64_t chunk = 1ull << chunk_log2;
  for (uint32_t c = wave; c < nchunks; c += nwaves) {
    const uint64_t base = (uint64_t)c << chunk_log2;
    const uint64_t len = (n - base < chunk) ? (n - base) : chunk;
    unsigned long long b = 0;
    for (uint64_t i = lane; i < len; i += 64) b += ll[in[base + i]];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) b += __shfl_xor(b, d, 64);
    if (lane == 0) bits[c] = b;
  }
}

// in-place exclusive scan of v[0..count), v[count] = total, *total_out = total.  One workgroup; thread t owns
// a run of consecutive elements, so there is a single block-wide scan whatever the count.
__global__ __launch_bounds__(1024) void k_scan(uint64_t* __restrict__ v, uint32_t count, uint64_t* __restrict__ total_out) {
  __shared__ unsigned long long wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint32_t ipt = (count + 1023u) / 1024u;
  const uint32_t lo = tid * ipt;
  const uint32_t hi = (lo + ipt < count) ? lo + ipt : count;
  unsigned long long mine = 0;
  // up to 8 elements per thread (K4: at most 8192 chunks) stay in registers: eight independent loads instead of
  // two passes of dependent ones (the kernel is pure latency: 11 us -> ~4 us)
  const bool small = ipt <= 8u;
  unsigned long long x8[8];
  if (small) {
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) {
      const uint32_t i = lo + k;
      x8[k] = (k < ipt && i < hi) ? v[i] : 0ull;
    }
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) mine += x8[k];
  } else {
    for (uint32_t i = lo; i < hi; ++i) mine += v[i];
  }
  unsigned long long s = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long t = __shfl_up(s, d, 64);
    if (lane >= d) s += t;
  }
  if (lane == 63) wsum[w] = s;
  __syncthreads();
  unsigned long long woff = 0, total = 0;
  for (int k = 0; k < 16; ++k) {
    const unsigned long long x = wsum[k];
    if (k < w) woff += x;
    total += x;
  }
  unsigned long long run = woff + s - mine;
  if (small) {
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) {
      const uint32_t i = lo + k;
      if (k < ipt && i < hi) v[i] = run;
      run += x8[k];
    }
  } else {
    for (uint32_t i = lo; i < hi; ++i) {
      const unsigned long long x = v[i];
      v[i] = run;
      run += x;
    }
  }
  if (tid == 0) {
    v[count] = total;
    if (total_out) *total_out = total;
  }
}

void launch_plan(const uint8_t* d_in, uint64_t n, uint32_t chunk_log2, uint32_t nchunks, const uint32_t* d_chunk_hist,
                 const ghf_code* d_code, uint64_t* d_chunk_off, uint64_t* d_total_bits, hipStream_t s) {
  uint32_t blocks = (nchunks + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  if (blocks == 0) blocks = 1;
  if (d_chunk_hist)
    hipLaunchKernelGGL(k_chunk_bits, dim3(blocks), dim3(256), 0, s, d_chunk_hist, nchunks, d_code, d_chunk_off);
  else
    hipLaunchKernelGGL(k_chunk_bits_direct, dim3(blocks), dim3(256), 0, s, d_in, n, chunk_log2, nchunks, d_code,
                       d_chunk_off);
  hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, d_chunk_off, nchunks, d_total_bits);
}

// ------------------------------------------------------------------------------------------------
// K7: decode.  k_build_decode_tables turns the header tables into left-justified first codes
// (FastCanonicalHuffDecoder, canonical_huff_encoder.cc:433-434) and a 2^lut_bits direct table
// {symbol, length} -- the reference's 8-bit length LUT (canonical_huff_encoder.cc:466-516) widened
// to min(max_len, 12) bits so that no linear extension is needed at any BASELINE config; longer
// codes fall back to the reference's linear search over first_code (cfind, canonical_huff_encoder.h:157-162).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_build_decode_tables(const ghf_code* __restrict__ code, DecTables* __restrict__ dt,
                                                             int* __restrict__ status) {
  __shared__ uint32_t fcl[36];
  __shared__ uint32_t sp[36];
  const int tid = threadIdx.x;
  const int max_len = code->max_len, min_len = code->min_len;
  if (max_len < 1 || max_len > 32 || min_len < 1 || min_len > max_len) {
    if (tid == 0) latch_status(status, GHF_E_FORMAT);
    return;
  }
  const int lb = max_len < kDecLutBitsMax ? max_len : kDecLutBitsMax;
  if (tid < 36) {
    uint32_t f = 0xFFFFFFFFu, p = 0;
    if (tid >= min_len && tid <= max_len) {
      f = code->first_code[tid] << (32 - tid);
      p = code->start_pos[tid];
    }
    fcl[tid] = f;
    sp[tid] = p;
    dt->fc_left[tid] = f;
    dt->start_pos[tid] = p;
  }
  for (int i = tid; i < GHF_NSYM; i += 256) dt->symbol[i] = (uint16_t)(code->symbol[i] > 256u ? 256u : code->symbol[i]);
  if (tid == 0) {
    dt->min_len = min_len;
    dt->max_len = max_len;
    dt->lut_bits = lb;
    dt->kind = 0;
    dt->root = 0;
  }
  if (tid < 16) dt->ticket[tid * 32] = 0;
  __syncthreads();
  for (uint32_t idx = tid; idx < (1u << lb); idx += 256) {
    const uint32_t v = idx << (32 - lb);
    uint16_t ent = 0;
    for (int len = min_len; len <= lb; ++len) {
      if (v >= fcl[len]) {
        const uint32_t k = sp[len] + ((v - fcl[len]) >> (32 - len));
        const uint32_t sym = k < GHF_NSYM ? code->symbol[k] : 256u;
        ent = (uint16_t)((sym > 256u ? 256u : sym) | ((uint32_t)len << 9));
        break;
      }
    }
    dt->lut[idx] = ent;
  }
  // two symbols per lookup when any two codes fit the index (small alphabets: 16-symbol data has max_len 5)
  const int pb = 2 * max_len <= kDecPairBitsMax ? 2 * max_len : 0;
  if (tid == 0) dt->pair_bits = pb;
  auto one = [&](uint32_t v) -> uint32_t {  // sym | len << 9 of the code at the top of v; 0: none
    for (int len = min_len; len <= max_len; ++len) {
      if (v >= fcl[len]) {
        const uint32_t k = sp[len] + ((v - fcl[len]) >> (32 - len));
        const uint32_t sym = k < GHF_NSYM ? code->symbol[k] : 256u;
        return (sym > 256u ? 256u : sym) | ((uint32_t)len << 9);
      }
    }
    return 0u;
  };
  for (uint32_t idx = tid; pb && idx < (1u << pb); idx += 256) {
    const uint32_t v = idx << (32 - pb);
    const uint32_t e0 = one(v);
    uint32_t ent = (1u << 30) | (1u << 16);  // not a data symbol: flagged, one bit consumed
    if (e0 && (e0 & 0x1FFu) != 256u) {
      const uint32_t l0 = e0 >> 9;
      const uint32_t e1 = one(v << l0);
      if (e1 && (e1 & 0x1FFu) != 256u) ent = (e0 & 0xFFu) | ((e1 & 0xFFu) << 8) | ((l0 + (e1 >> 9)) << 16);
      else ent = (1u << 30) | (l0 << 16);
    }
    dt->lut2[idx] = ent;
  }
}

// .crs: the same direct table, filled by walking the tree DecodeHuffTree::do_build_tree would rebuild
// (include/huff_tree.cc:289-303); what the table cannot resolve is walked bit by bit like decode_byte does (:255-271).
__global__ __launch_bounds__(256) void k_crs_decode_tables(const ghf_tree* __restrict__ tree, DecTables* __restrict__ dt,
                                                           int* __restrict__ status) {
  __shared__ uint16_t tl[256], tr[256];
  __shared__ uint32_t s_min;
  const int tid = threadIdx.x;
  const int max_len = (int)tree->max_len;
  const uint32_t root = tree->root, nl = tree->n_leaves;
  if (max_len < 1 || max_len > 32 || nl < 2 || nl > 256 || root < 256 || root >= 256 + nl - 1) {
    if (tid == 0) latch_status(status, GHF_E_FORMAT);
    return;
  }
  tl[tid] = tree->left[tid];
  tr[tid] = tree->right[tid];
  if (tid == 0) s_min = 64;
  __syncthreads();
  const int lb = max_len < kDecLutBitsMax ? max_len : kDecLutBitsMax;
  dt->tl[tid] = tl[tid];
  dt->tr[tid] = tr[tid];
  if (tid < 36) {
    dt->fc_left[tid] = 0xFFFFFFFFu;
    dt->start_pos[tid] = 0;
  }
  for (int i = tid; i < GHF_NSYM; i += 256) dt->symbol[i] = 256;
  if (tid < 16) dt->ticket[tid * 32] = 0;
  uint32_t mn = 64;
  for (uint32_t idx = tid; idx < (1u << lb); idx += 256) {
    uint32_t node = root;
    uint16_t ent = 0;
    for (int l = 1; l <= lb; ++l) {
      const uint32_t p = node - 256u;
      if (p >= nl - 1) break;  // a child id that is neither a leaf nor one of the nl - 1 parents: malformed, entry stays 0
      node = ((idx >> (lb - l)) & 1u) ? tr[p] : tl[p];
      if (node < 256u) {
        ent = (uint16_t)(node | ((uint32_t)l << 9));
        mn = (uint32_t)l < mn ? (uint32_t)l : mn;
        break;
      }
    }
    dt->lut[idx] = ent;
  }
  atomicMin(&s_min, mn);
  __syncthreads();
  if (tid == 0) {
    dt->min_len = (int32_t)(s_min <= (uint32_t)lb ? s_min : (uint32_t)lb);
    dt->max_len = max_len;
    dt->lut_bits = lb;
    dt->pair_bits = 0;
    dt->kind = 1;
    dt->root = root;
  }
}

void launch_crs_decode_tables(const ghf_tree* d_tree, DecTables* d_dt, int* d_status, hipStream_t s) {
  hipLaunchKernelGGL(k_crs_decode_tables, dim3(1), dim3(256), 0, s, d_tree, d_dt, d_status);
}

void launch_build_decode_tables(const ghf_code* d_code, DecTables* d_dt, int* d_status, hipStream_t s) {
  hipLaunchKernelGGL(k_build_decode_tables, dim3(1), dim3(256), 0, s, d_code, d_dt, d_status);
}

struct DecLds {  // K6 (side-car reconstruction): 4 waves, plain table
  alignas(16) uint32_t in[kDecWaves][kDecInWords + 4];
  alignas(16) uint16_t lut[1 << kDecLutBitsMax];
  uint32_t fcl[36];
  uint32_t sp[36];
  uint16_t symbol[GHF_NSYM + 3];
  uint16_t tl[256], tr[256];  // kind 1 (.crs): the tree
  uint32_t root;
  int kind;
  int status0;
};

// K7: the direct table is REPLICATED so that the 64 random lookups of a wave do not pile up on a few LDS banks
// (PMC, 256 MiB uniform: 77 % of all LDS cycles of the un-replicated kernel were bank-conflict cycles).  The
// table gets a fixed 32 KiB; with lut_bits index bits there is room for R = 2^(14 - lut_bits) copies, lane l uses
// copy l % R:  9-bit codes (uniform bytes) -> 32 copies, every lane of a 32-lane LDS group in its own bank pair;
// 12-bit tables -> 4 copies (skewed data hits few, mostly identical entries anyway: identical addresses broadcast).
constexpr int kDec7Threads = 1024;
constexpr int kDec7Waves = kDec7Threads / kWave;
constexpr int kDec7LutLog2 = 13;
constexpr int kDec7InBytes = 4608;  // staged span per wave: 4096 symbols at <= 9 bits average (a byte-Huffman code averages <= 8.1)
constexpr int kDec7InWords = kDec7InBytes / 4;
struct DecLds7 {
  alignas(16) uint32_t in[kDec7Waves][kDec7InWords + 4];  // compressed span of the wave's group, big-endian words
  alignas(16) uint32_t out[kDec7Waves][1024];             // the group's 4096 output bytes, lane-major, before the copy-out
  alignas(16) uint16_t lut[1 << kDec7LutLog2];
  uint32_t fcl[36];
  uint32_t sp[36];
  uint16_t symbol[GHF_NSYM + 3];
  uint16_t tl[256], tr[256];  // kind 1 (.crs): the tree
  uint32_t root;
  int kind;
  int status0;
};
static_assert(sizeof(DecLds7) <= 160 * 1024, "one workgroup of 16 waves per CU");

template <typename LT>
__device__ __forceinline__ void dec_small_load(LT& L, const DecTables* dt, int tid, int nthreads) {
  if (tid < 36) {
    L.fcl[tid] = dt->fc_left[tid];
    L.sp[tid] = dt->start_pos[tid];
  }
  for (int i = tid; i < GHF_NSYM; i += nthreads) L.symbol[i] = dt->symbol[i];
  for (int i = tid; i < 256; i += nthreads) {
    L.tl[i] = dt->tl[i];
    L.tr[i] = dt->tr[i];
  }
  if (tid == 0) {
    L.kind = dt->kind;
    L.root = dt->root;
  }
}

__device__ __forceinline__ void dec_lds_load(DecLds& L, const DecTables* dt, int tid, int nthreads) {
  const int lut_bits = dt->lut_bits;
  const uint4* src = reinterpret_cast<const uint4*>(dt->lut);
  uint4* dst = reinterpret_cast<uint4*>(L.lut);
  for (int i = tid; i < ((1 << lut_bits) * 2 + 15) / 16; i += nthreads) dst[i] = src[i];
  dec_small_load(L, dt, tid, nthreads);
}

// replicated fill: entry idx, copy r at lut[(idx << rshift) | r].  With a pair table the 32 KiB are split:
// lower half = lut2 (4096 u32 slots), upper half = the one-symbol table (8192 u16 slots, used for ragged tails
// and the end mark).
__device__ __forceinline__ void dec_lds_load7(DecLds7& L, const DecTables* dt, int tid, int nthreads) {
  const int pb = dt->pair_bits;
  const int slots_log2 = pb ? kDec7LutLog2 - 1 : kDec7LutLog2;
  const int rshift = slots_log2 - dt->lut_bits;
  uint32_t* dst = reinterpret_cast<uint32_t*>(L.lut + (pb ? (1 << (kDec7LutLog2 - 1)) : 0));
  for (int i = tid; i < (1 << (slots_log2 - 1)); i += nthreads) {  // two u16 slots per store
    const uint32_t a = dt->lut[(2 * i) >> rshift], b = dt->lut[(2 * i + 1) >> rshift];
    dst[i] = a | (b << 16);
  }
  if (pb) {
    uint32_t* d2 = reinterpret_cast<uint32_t*>(L.lut);
    const int r2 = kDecPairBitsMax - pb;
    for (int i = tid; i < (1 << kDecPairBitsMax); i += nthreads) d2[i] = dt->lut2[i >> r2];
  }
  dec_small_load(L, dt, tid, nthreads);
}

// codes longer than the direct table: the reference's linear extension (canonical_huff_encoder.cc:554-557).
// returns sym | len << 16
template <typename LT>
__device__ __forceinline__ uint32_t dec_long(const LT& L, uint32_t hi, int lut_bits, int max_len) {
  if (L.kind == 1) {  // .crs: walk the tree from the root (huff_tree.cc:255-271); malformed trees end in "no symbol"
    uint32_t node = L.root;
    for (int l = 1; l <= max_len; ++l) {
      const uint32_t p = node - 256u;
      if (p >= 256u) break;
      node = ((hi >> (32 - l)) & 1u) ? L.tr[p] : L.tl[p];
      if (node < 256u) return node | ((uint32_t)l << 16);
    }
    return 256u | ((uint32_t)max_len << 16);
  }
  int l = lut_bits + 1;
  while (l < max_len && hi < L.fcl[l]) ++l;
  const uint32_t k = L.sp[l] + ((hi - L.fcl[l]) >> (32 - l));
  return (k < GHF_NSYM ? (uint32_t)L.symbol[k] : 256u) | ((uint32_t)l << 16);
}

template <bool STAGED>
struct DecIn {
  const uint32_t* in;   // staged big-endian words
  const uint8_t* src;   // unstaged: raw bytes of the span
  uint64_t span;
  __device__ __forceinline__ uint32_t fetch(uint32_t widx) const {
    if (STAGED) return in[widx];
    const uint64_t b = (uint64_t)widx * 4;
    uint32_t r = 0;
    for (int k = 0; k < 4; ++k) r = (r << 8) | (b + k < span ? (uint32_t)src[b + k] : 0u);
    return r;
  }
};

// Decode this lane's segment (cnt symbols starting at bit `pos` of the span) and store the bytes.
// The window W holds 64 stream bits, `o` of them (from the top) already consumed; one symbol costs a 64-bit
// shift, the table lookup and an add.  K symbols are decoded between two refill checks -- the caller picks
// K = 32 / max_len (<= 4), which keeps o + max_len <= 64 at every lookup -- and LONG says whether codes longer
// than the direct table exist at all.  Returns an accumulator whose bit 8 is set when something is wrong.
#if defined(GHF_EXP) && GHF_EXP == 1
#define GHF_EXP_LUT(real, v) (((v) >> 24) | ((8u + ((v) >> 31)) << 9))   /* experiment: no table lookup */
#else
#define GHF_EXP_LUT(real, v) (real)
#endif
// LDSOUT: the 16-byte pieces go to the wave's LDS tile (`outl` = this lane's 64-byte row, pieces XOR-swizzled by
// `osw` so that the 16 lanes of a write phase hit 16 different bank groups); the caller copies the tile out.
template <bool STAGED, int K, bool LONG, bool PAIR, bool LDSOUT, typename LT>
__device__ __forceinline__ uint32_t decode_segment(const LT& L, const DecIn<STAGED>& I, int lut_bits, int max_len,
                                                   uint64_t pos, uint32_t cnt, bool valid, bool all_full, uint8_t* optr,
                                                   int has_next, uint64_t expect_bits, int rshift, uint32_t rep,
                                                   const uint16_t* lut1, int pair_bits, uint32_t rep2, uint32_t* outl,
                                                   uint32_t osw) {
  const int lsh = 32 - lut_bits;
  const char* lut1b = reinterpret_cast<const char*>(lut1 + rep);  // this lane's replica: address = base + (index << (rshift + 1))
  uint32_t widx = (uint32_t)(pos >> 5);
  uint32_t o = (uint32_t)(pos & 31u);
  const uint32_t o0 = o, widx0 = widx;
  uint64_t W = ((uint64_t)I.fetch(widx) << 32) | I.fetch(widx + 1);
  uint32_t nextw = I.fetch(widx + 2);
  widx += 3;
  uint32_t bad_acc = 0;

#if defined(GHF_EXP) && GHF_EXP == 3
#define GHF_REFILL()                 \
  if (o >= 32u) {                    \
    W = (W << 32) | nextw;           \
    o -= 32u;                        \
    nextw = nextw * 2654435761u + widx++; \
  }
#else
#define GHF_REFILL()                 \
  if (o >= 32u) {                    \
    W = (W << 32) | nextw;           \
    o -= 32u;                        \
    nextw = I.fetch(widx++);         \
  }
#endif
#define GHF_DEC(ENT)                                          \
  do {                                                        \
    const uint32_t v_ = (uint32_t)((W << o) >> 32);           \
    ENT = GHF_EXP_LUT(*reinterpret_cast<const uint16_t*>(lut1b + ((v_ >> lsh) << (rshift + 1))), v_); \
    if (LONG && __builtin_expect((ENT >> 9) == 0, 0)) {       \
      const uint32_t r_ = dec_long(L, v_, lut_bits, max_len); \
      ENT = (r_ & 0x1FFu) | ((r_ >> 16) << 9);                \
    }                                                         \
    o += ENT >> 9;                                            \
  } while (0)

  if (STAGED && PAIR && all_full) {
    // small alphabet: any two codes fit lut2's index, so one lookup yields two symbols and the serial
    // shift -> lookup -> add chain is half as long.  Two pairs (<= 2 * pair_bits <= 24 bits) per refill check.
    const uint32_t* lut2 = reinterpret_cast<const uint32_t*>(L.lut);
    const int psh = 32 - pair_bits, r2 = kDecPairBitsMax - pair_bits;
    const char* lut2b = reinterpret_cast<const char*>(lut2 + rep2);
    uint32_t bad2 = 0;
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      uint32_t wq[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#if defined(GHF_EXP) && (GHF_EXP == 8 || GHF_EXP == 9)
        wq[k] = W + k + q; continue;
#endif
        GHF_REFILL();
        const uint32_t va = (uint32_t)((W << o) >> 32);
        const uint32_t ea = *reinterpret_cast<const uint32_t*>(lut2b + ((va >> psh) << (r2 + 2)));
        o += (ea >> 16) & 31u;
        const uint32_t vb = (uint32_t)((W << o) >> 32);
        const uint32_t eb = *reinterpret_cast<const uint32_t*>(lut2b + ((vb >> psh) << (r2 + 2)));
        o += (eb >> 16) & 31u;
        bad2 |= ea | eb;
        wq[k] = __builtin_amdgcn_perm(eb, ea, 0x05040100u);  // {ea.sym0, ea.sym1, eb.sym0, eb.sym1}
      }
      if (LDSOUT) *reinterpret_cast<uint4*>(outl + (((uint32_t)q ^ osw) << 2)) = make_uint4(wq[0], wq[1], wq[2], wq[3]);
      else if (valid) *reinterpret_cast<uint4*>(optr + q * 16) = make_uint4(wq[0], wq[1], wq[2], wq[3]);
    }
    bad_acc |= (bad2 >> 22) & 256u;
  } else if (STAGED && all_full) {
    // full segments everywhere (all groups but the stream's last): 16 output bytes per store
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      uint32_t wq[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#if defined(GHF_EXP) && (GHF_EXP == 8 || GHF_EXP == 9)
        wq[k] = W + k + q; continue;
#endif
        uint32_t e[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          if ((4 * k + jj) % K == 0) GHF_REFILL();
          GHF_DEC(e[jj]);
        }
        bad_acc |= e[0] | e[1] | e[2] | e[3];
        // byte 0 of four entries -> one dword (v_perm_b32): {a.b0, b.b0} then {lo16, hi16}
        const uint32_t lo = __builtin_amdgcn_perm(e[1], e[0], 0x0C0C0400u);
        const uint32_t hi = __builtin_amdgcn_perm(e[3], e[2], 0x0C0C0400u);
        wq[k] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
      }
      if (LDSOUT) *reinterpret_cast<uint4*>(outl + (((uint32_t)q ^ osw) << 2)) = make_uint4(wq[0], wq[1], wq[2], wq[3]);
      else if (valid) *reinterpret_cast<uint4*>(optr + q * 16) = make_uint4(wq[0], wq[1], wq[2], wq[3]);
    }
  } else if (valid) {
    for (uint32_t i = 0; i < cnt; ++i) {
      uint32_t e;
      GHF_REFILL();
      GHF_DEC(e);
      bad_acc |= e;
      optr[i] = (uint8_t)e;
    }
  }
  if (valid) {
    // the index says where the next segment starts: an end-to-end check of every segment
    if (has_next == 1) {
      const uint64_t used = (uint64_t)(widx - widx0 - 3) * 32 + o - o0;
      if (used != expect_bits) bad_acc |= 256u;
    } else if (has_next == 0) {
      uint32_t e;
      GHF_REFILL();
      GHF_DEC(e);
      if ((e & 0x1FFu) != 256u) bad_acc |= 256u;  // canonical_huff_encoder.cc:404: the end mark must follow
    }
  }
#undef GHF_DEC
#undef GHF_REFILL
  return bad_acc & 256u;
}

// K7.  Persistent waves; each pass a wave takes 64 consecutive segments (4096 symbols):
//   1. the compressed span of those segments (known from the side-car) is copied into LDS with
//      coalesced 16-byte loads, byte-swapped to big-endian words;
//   2. every lane decodes its 64 symbols from a 64-bit window: one LDS table lookup per symbol;
//   3. every 16 symbols the lane stores 16 output bytes straight to HBM (its 64 bytes are contiguous).
// The loop is software-pipelined over groups so that no HBM latency is exposed: while group i is decoded,
// the span of group i+1 is in flight into registers and the side-car entries of group i+2 are in flight too.
struct DecGroup {      // what a lane knows about its segment in one group (all per-lane unless noted)
  uint64_t sbit, nbit; // start bit of my segment / of the next one
  uint64_t byte0;      // uniform: first staged byte (16-aligned)
  uint64_t span;       // uniform: staged bytes
};

// Side-car words of a lane's segment and of the one after it, as loaded: the loads are issued a whole pass before
// dec_meta_bits() combines them, and nothing in between may need their values (a single dependent use right behind
// the loads makes the compiler wait for every older memory operation, including the span prefetch).
struct DecMeta {
  uint64_t cb0, cb1;
  uint32_t sb0, sb1;
};

__device__ __forceinline__ void dec_issue_meta(const DecParams& P, uint64_t group, int lane, DecMeta& M) {
  const uint64_t last = P.n_segs - 1;
  const uint64_t seg = group * 64 + lane;
  const uint64_t s0 = seg < last ? seg : last, s1 = seg + 1 < last ? seg + 1 : last;  // clamped: unconditional loads
  M.cb0 = P.chunk_bit[(s0 * kSegSymbols) >> P.chunk_log2];
  M.sb0 = P.seg_bit[s0];
  M.cb1 = P.chunk_bit[(s1 * kSegSymbols) >> P.chunk_log2];
  M.sb1 = P.seg_bit[s1];
}

__device__ __forceinline__ void dec_meta_bits(const DecParams& P, uint64_t group, int lane, const DecMeta& M, uint64_t& sbit,
                                              uint64_t& nbit) {
  const uint64_t stream_end_bit = P.stream_bytes * 8;
  const uint64_t seg = group * 64 + lane;
  sbit = seg < P.n_segs ? M.cb0 + M.sb0 : stream_end_bit;
  nbit = seg + 1 < P.n_segs ? M.cb1 + M.sb1 : stream_end_bit;
}

__device__ __forceinline__ void dec_span(const DecParams& P, uint64_t group, int lane, int max_len, DecGroup& G) {
  const uint64_t stream_end_bit = P.stream_bytes * 8;
  const uint64_t seg0 = group * 64;
  const bool valid = seg0 + lane < P.n_segs;
  const uint64_t B0 = __shfl(G.sbit, 0, 64);
  uint64_t B1 = __shfl(G.nbit, 63, 64);
  if (seg0 + 64 >= P.n_segs) {
    // last group: nothing tells where it ends; bound it by its last segment's worst case (+ end mark)
    uint64_t m = valid ? G.sbit + 65ull * (uint64_t)max_len : 0ull;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const uint64_t o = __shfl_xor(m, d, 64);
      m = o > m ? o : m;
    }
    B1 = m < stream_end_bit ? m : stream_end_bit;
  }
  G.byte0 = (B0 >> 3) & ~15ull;
  uint64_t byte1 = ((B1 + 7) >> 3) + 12;  // window look-ahead
  if (byte1 > P.stream_bytes) byte1 = P.stream_bytes;
  if (G.byte0 > byte1) G.byte0 = byte1 & ~15ull;  // corrupt side-car: caught by the checks below
  G.span = byte1 - G.byte0;
}

constexpr int kDecVec = (kDec7InBytes + 1023) / 1024;  // 16-byte vectors per lane that cover a staged span

__global__ __launch_bounds__(kDec7Threads, 4) void k_decode(DecParams P) {
  __shared__ DecLds7 L;
  const int tid = threadIdx.x;
  if (tid == 0) L.status0 = *P.status;  // one read per workgroup: the exit must be uniform
  __syncthreads();
  if (L.status0 != 0) return;
  const int lut_bits = P.dt->lut_bits, max_len = P.dt->max_len;
  dec_lds_load7(L, P.dt, tid, kDec7Threads);
  __syncthreads();
  const int pair_bits = P.dt->pair_bits;
  const int rshift = (pair_bits ? kDec7LutLog2 - 1 : kDec7LutLog2) - lut_bits;
  const uint32_t rep = (uint32_t)tid & ((1u << rshift) - 1u);
  const uint32_t rep2 = (uint32_t)tid & ((1u << (kDecPairBitsMax - pair_bits)) - 1u);
  const uint16_t* lut1 = L.lut + (pair_bits ? (1 << (kDec7LutLog2 - 1)) : 0);
  if (blockIdx.x == 0 && tid == 0 && P.out_bytes) *P.out_bytes = P.n_symbols;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform -> scalar loop control
  const uint64_t ngroups = (P.n_segs + 63) >> 6;
  const uint64_t stream_end_bit = P.stream_bytes * 8;
  const uint64_t full_bytes = P.stream_bytes & ~15ull;  // whole 16-byte vectors of the stream
  const bool out_aligned = (((uintptr_t)P.out) & 15u) == 0;
  uint32_t* in = L.in[wave];
  uint32_t bad_acc = 0;
  // groups are handed out by a global ticket counter, not by a fixed stride: a wave that starts late (e.g. because
  // another kernel occupied its CU) simply takes fewer groups instead of becoming the kernel's straggler
  const uint32_t ncls = gridDim.x < 16u ? gridDim.x : 16u;  // every class needs at least one workgroup
  const uint32_t cls = blockIdx.x % ncls;
  auto ticket_issue = [&]() -> unsigned int {  // the atomic's return value stays in a VGPR until ticket_group() needs it
    unsigned int t = 0;
    if (lane == 0) t = atomicAdd(&P.dt->ticket[cls * 32], 1u);
    return t;
  };
  // every wave's first three groups are fixed (no round trip to a counter before the first load can be issued: three
  // dependent atomics on 16 counters cost the 4096 waves several microseconds of start-up); tickets number the rest
  const uint64_t nwaves = (uint64_t)gridDim.x * kDec7Waves;
  const uint64_t wid = (uint64_t)blockIdx.x * kDec7Waves + (uint64_t)wave;
  auto ticket_group = [&](unsigned int t) -> uint64_t {
    return 3 * nwaves + (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)t) * ncls + cls;
  };
  uint64_t group = wid;
  if (group >= ngroups) return;
  uint64_t g1 = wid + nwaves, g2 = wid + 2 * nwaves;  // this wave's next two groups
  const uint64_t glast = ngroups - 1;
  auto clampg = [&](uint64_t g) { return g < ngroups ? g : glast; };  // past the end: redundant, harmless loads

  // issue the 16-byte loads of a group's span (vector k of this lane = bytes byte0 + k*1024 + lane*16 ..)
  auto issue = [&](const DecGroup& G, uint4 (&R)[kDecVec]) {
#pragma unroll
    for (int k = 0; k < kDecVec; ++k) {
      const uint64_t o = (uint64_t)k * 1024 + (uint64_t)lane * 16;
      const bool ok = o < G.span && G.byte0 + o + 16 <= full_bytes;  // lanes behind the span re-read byte 0 (an L2 hit)
      R[k] = *reinterpret_cast<const uint4*>(P.stream + (ok ? G.byte0 + o : 0));
    }
  };

  DecGroup cur, nxt;
  uint4 R[kDecVec];
  DecMeta M;  // raw side-car words of the group after `cur` (of the one after that once the pass has issued its loads)
  dec_issue_meta(P, group, lane, M);
  dec_meta_bits(P, group, lane, M, cur.sbit, cur.nbit);
  dec_span(P, group, lane, max_len, cur);
  issue(cur, R);
  dec_issue_meta(P, clampg(g1), lane, M);

  // One pass over a group.  HOT = the group is complete, staged, plausible and not the stream's last: the body then has
  // no data-dependent branch around its memory operations, so the compiler can count them -- the wait for the
  // prefetched span becomes "all but the youngest four" (this group's output stores) instead of vmcnt(0), and the wave
  // no longer sleeps until its own stores are acknowledged by L2 (which is what bounded this kernel before).
  auto pass = [&](auto hot_tag) {
    constexpr bool HOT = decltype(hot_tag)::value;
    // ---- 1. this group's span: registers -> LDS (big-endian words); everything behind it reads as zero
    wave_sync();
#pragma unroll
    for (int k = 0; k < kDecVec; ++k) {
      const uint64_t o = (uint64_t)k * 1024 + (uint64_t)lane * 16;
      const bool ok = (o < cur.span) && (cur.byte0 + o + 16 <= full_bytes);
      uint4 v = R[k];
      v = ok ? make_uint4(bswap32(v.x), bswap32(v.y), bswap32(v.z), bswap32(v.w)) : make_uint4(0, 0, 0, 0);
      if ((k + 1) * 1024 <= kDec7InBytes || o < (uint64_t)kDec7InBytes) *reinterpret_cast<uint4*>(in + (o >> 2)) = v;
    }
    if (lane < 4) in[kDec7InWords + lane] = 0;
    if (!HOT && cur.byte0 + cur.span > full_bytes && full_bytes >= cur.byte0 && lane == 0) {
      // the stream's last, incomplete 16 bytes: byte loads, never past the end of the buffer
      uint32_t q[4] = {0, 0, 0, 0};
      for (uint64_t j = 0; full_bytes + j < P.stream_bytes; ++j) q[j >> 2] |= (uint32_t)P.stream[full_bytes + j] << (24 - 8 * (j & 3));
      const uint64_t w = (full_bytes - cur.byte0) >> 2;
      if (w + 3 < (uint64_t)kDec7InWords + 4) {
        in[w] = q[0]; in[w + 1] = q[1]; in[w + 2] = q[2]; in[w + 3] = q[3];
      }
    }
    wave_sync();
    // ---- prefetch: span of the next group (its side-car entries arrived during the last decode), side-car of the one after
    dec_meta_bits(P, clampg(g1), lane, M, nxt.sbit, nxt.nbit);  // loaded a pass ago
    dec_span(P, clampg(g1), lane, max_len, nxt);
    issue(nxt, R);
    dec_issue_meta(P, clampg(g2), lane, M);
    const unsigned int t3 = ticket_issue();  // resolved after the decode
    // ---- 2./3. decode
    const uint64_t seg0 = group * 64;
    const uint64_t seg = seg0 + lane;
    const uint64_t sym0 = seg * kSegSymbols;
    const uint64_t pos = cur.sbit - cur.byte0 * 8;
    uint8_t* optr = P.out + sym0;
    const uint64_t expect = cur.nbit - cur.sbit;
    const uint8_t* src = P.stream + cur.byte0;
    uint32_t* outl = L.out[wave] + lane * 16;
    const uint32_t osw = ((uint32_t)lane >> 2) & 3u;
#define GHF_SEG_ARGS L, I, lut_bits, max_len, pos, cnt, valid, all_full, optr, has_next, expect, rshift, rep, lut1, pair_bits, rep2, outl, osw
    if (HOT) {
      const uint32_t cnt = kSegSymbols;
      const bool valid = true, all_full = true;
      const int has_next = 1;
      DecIn<true> I{in, src, cur.span};
      // K = 32 / max_len symbols per refill check; LONG = codes beyond the direct table exist
      if (pair_bits) bad_acc |= decode_segment<true, 4, false, true, true>(GHF_SEG_ARGS);
      else if (max_len <= 8) bad_acc |= decode_segment<true, 4, false, false, true>(GHF_SEG_ARGS);
      else if (max_len <= 10) bad_acc |= decode_segment<true, 3, false, false, true>(GHF_SEG_ARGS);
      else if (max_len <= kDecLutBitsMax) bad_acc |= decode_segment<true, 2, false, false, true>(GHF_SEG_ARGS);
      else if (max_len <= 16) bad_acc |= decode_segment<true, 2, true, false, true>(GHF_SEG_ARGS);
      else bad_acc |= decode_segment<true, 1, true, false, true>(GHF_SEG_ARGS);
      // copy-out: four fully coalesced 1 KiB stores per wave, straight-line, so that the compiler can count them and
      // the next pass waits for "all but the youngest four" memory operations instead of for everything (the stores
      // used to sit in the rolled decode loop: every pass slept until L2 had acknowledged its own stores, and 16-byte
      // pieces at a 64-byte stride reached HBM as 1.31x the output bytes)
      wave_sync();
      {
        const uint32_t* ot = L.out[wave];
        uint8_t* og = P.out + seg0 * kSegSymbols + (uint32_t)lane * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t sl = (uint32_t)r * 16 + ((uint32_t)lane >> 2);        // the lane whose row holds my piece
          const uint32_t piece = ((uint32_t)lane & 3u) ^ ((sl >> 2) & 3u);
          const uint4 v = *reinterpret_cast<const uint4*>(ot + sl * 16 + piece * 4);
#if defined(GHF_EXP) && (GHF_EXP == 2 || GHF_EXP == 9)
          if (v.x == 0x12345678u && v.y == 0x9abcdef0u)  // experiment: no output stores
#endif
          *reinterpret_cast<uint4*>(og + r * 1024) = v;
        }
      }
    } else {
      const bool valid = seg < P.n_segs;
      const bool bad = valid && (cur.sbit >= stream_end_bit || cur.nbit > stream_end_bit || cur.nbit < cur.sbit ||
                                 cur.sbit < cur.byte0 * 8);
      if (__ballot(bad)) {
        if (bad) latch_status(P.status, GHF_E_CORRUPT);
      } else {
        const bool staged = cur.span <= (uint64_t)kDec7InBytes;
        uint32_t cnt = 0;
        if (valid) cnt = (P.n_symbols - sym0 >= (uint64_t)kSegSymbols) ? (uint32_t)kSegSymbols : (uint32_t)(P.n_symbols - sym0);
        const bool all_full = (__ballot(valid && cnt != (uint32_t)kSegSymbols) == 0) && out_aligned;
        // 1: the side-car says where the next segment starts; 0: the end mark must follow; 2: nothing to check
        const int has_next = seg + 1 < P.n_segs ? 1 : (P.no_end_mark ? 2 : 0);
        if (staged) {
          DecIn<true> I{in, src, cur.span};
          if (max_len <= 8) bad_acc |= decode_segment<true, 4, false, false, false>(GHF_SEG_ARGS);
          else if (max_len <= kDecLutBitsMax) bad_acc |= decode_segment<true, 2, false, false, false>(GHF_SEG_ARGS);
          else bad_acc |= decode_segment<true, 1, true, false, false>(GHF_SEG_ARGS);
        } else {
          DecIn<false> I{in, src, cur.span};
          bad_acc |= decode_segment<false, 1, true, false, false>(GHF_SEG_ARGS);
        }
      }
    }
#undef GHF_SEG_ARGS
    // ---- rotate (all of these values have long arrived)
    cur = nxt;
    group = g1;
    g1 = g2;
    g2 = ticket_group(t3);
  };
  auto is_hot = [&]() -> bool {  // wave-uniform
    if (group + 1 >= ngroups || !out_aligned) return false;                   // the last group may be ragged / carries the end mark
    if (cur.span > (uint64_t)kDec7InBytes || cur.byte0 + cur.span > full_bytes) return false;
    const bool bad = cur.sbit >= stream_end_bit || cur.nbit > stream_end_bit || cur.nbit < cur.sbit || cur.sbit < cur.byte0 * 8;
    return __ballot(bad) == 0;
  };
  while (group < ngroups) {
    if (is_hot()) {
      // drain once on entry: the hot loop's waits are then computed from its own back edge alone (exact counts)
      // instead of being merged with whatever the cold paths left outstanding
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
      do pass(std::true_type{});
      while (is_hot());
    }
    if (group < ngroups) pass(std::false_type{});
  }
#ifndef GHF_EXP
  if (bad_acc & 256u) latch_status(P.status, GHF_E_CORRUPT);  // a data symbol can never be 256
#else
  if (bad_acc == 0xFFFFFFFFu) latch_status(P.status, GHF_E_CORRUPT);
#endif
}

void launch_decode(const DecParams& p, hipStream_t s) {
  const uint64_t groups = (p.n_segs + 63) / 64;
  uint64_t blocks = (groups + kDec7Waves - 1) / kDec7Waves;
  if (blocks == 0) return;
  if (blocks > 256) blocks = 256;  // persistent: one workgroup of 16 waves per CU (its LDS tiles + table take 155 KiB)
  hipLaunchKernelGGL(k_decode, dim3((uint32_t)blocks), dim3(kDec7Threads), 0, s, p);
}

// ------------------------------------------------------------------------------------------------
// K6: rebuild the side-car of a FOREIGN stream (a .crs2 written by the reference has no sync points).
// Huffman codes self-synchronise: a decoder started at a wrong bit falls into step with the true
// code boundaries after a few symbols.  The body is cut into 512-bit subsequences; every thread decodes
// its subsequence from its current guess of the first code boundary and tells its right neighbour where
// it landed.  Thread 0 starts at a true boundary, so the fixed point of this iteration is the true
// segmentation; passes repeat (only threads whose guess changed redo work) until nothing changes.
// Then symbol counts are prefix-summed, the end mark fixes n, and one more pass writes the bit position
// of every 64th symbol -- the same side-car K5 emits.
// ------------------------------------------------------------------------------------------------
constexpr int kSubBits = 512;

struct BitReader {  // left-justified 64-bit window over big-endian words (LDS)
  const uint32_t* in;
  uint32_t widx;
  uint64_t window;
  int avail;
  __device__ __forceinline__ void init(const uint32_t* words, uint64_t pos) {
    in = words;
    widx = (uint32_t)(pos >> 5);
    const uint32_t off = (uint32_t)(pos & 31u);
    window = (((uint64_t)in[widx] << 32) | in[widx + 1]) << off;
    widx += 2;
    avail = 64 - (int)off;
  }
  __device__ __forceinline__ uint32_t hi() {
    if (avail < 32) {
      window |= (uint64_t)in[widx++] << (32 - avail);
      avail += 32;
    }
    return (uint32_t)(window >> 32);
  }
  __device__ __forceinline__ void skip(uint32_t len) {
    window <<= len;
    avail -= (int)len;
  }
};

__device__ __forceinline__ uint32_t dec_any(const DecLds& L, uint32_t hi, int lut_bits, int max_len, uint32_t& len) {
  const uint32_t ent = L.lut[hi >> (32 - lut_bits)];
  len = ent >> 9;
  if (len) return ent & 0x1FFu;
  const uint32_t r = dec_long(L, hi, lut_bits, max_len);
  len = r >> 16;
  return r & 0xFFFFu;
}

// stage the bits of 64 consecutive subsequences (+ look-ahead) of the body into this wave's LDS words;
// returns the bit offset of subsequence `sub0` inside the staged words
__device__ __forceinline__ uint64_t stage_subs(const SyncParams& P, uint64_t sub0, uint32_t* in, int lane) {
  const uint64_t bit0 = P.body_bit0 + sub0 * kSubBits;
  const uint64_t byte0 = (bit0 >> 3) & ~15ull;
  uint64_t byte1 = ((bit0 + 64ull * kSubBits + 7) >> 3) + 16;
  if (byte1 > P.stream_bytes) byte1 = P.stream_bytes;
  const uint64_t span = byte1 > byte0 ? byte1 - byte0 : 0;
  const uint8_t* src = P.stream + byte0;
  for (uint64_t o = (uint64_t)lane * 16; o < span; o += 1024) {
    uint4 v;
    if (o + 16 <= span) {
      v = *reinterpret_cast<const uint4*>(src + o);
    } else {
      uint32_t q[4] = {0, 0, 0, 0};
      for (uint32_t j = 0; o + j < span; ++j) q[j >> 2] |= (uint32_t)src[o + j] << (8 * (j & 3));
      v = make_uint4(q[0], q[1], q[2], q[3]);
    }
    *reinterpret_cast<uint4*>(in + (o >> 2)) = make_uint4(bswap32(v.x), bswap32(v.y), bswap32(v.z), bswap32(v.w));
  }
  const uint32_t wend = (uint32_t)((span + 15) >> 4) << 2;
  for (uint32_t k = wend + lane; k < (uint32_t)kDecInWords + 4; k += 64) in[k] = 0;
  return bit0 - byte0 * 8;
}

__global__ __launch_bounds__(kDecThreads) void k_sync_pass(SyncParams P) {
  __shared__ DecLds L;
  const int tid = threadIdx.x;
  dec_lds_load(L, P.dt, tid, kDecThreads);
  __syncthreads();
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lut_bits = P.dt->lut_bits, max_len = P.dt->max_len;
  const uint64_t ngroups = (P.nsub + 63) >> 6;
  const uint64_t body_bits = P.end_bit - P.body_bit0;
  uint32_t* in = L.in[wave];
  for (uint64_t g = (uint64_t)blockIdx.x * kDecWaves + wave; g < ngroups; g += (uint64_t)gridDim.x * kDecWaves) {
    const uint64_t sub = g * 64 + lane;
    const bool valid = sub < P.nsub;
    uint32_t st = 0;
    bool work = false;
    if (valid) {
      st = P.start[sub];
      work = P.used[sub] != st;
    }
    if (!__ballot(work)) continue;  // the whole wave's results are still current
    wave_sync();
    const uint64_t base = stage_subs(P, g * 64, in, lane);
    wave_sync();
    if (work) {
      const uint64_t sub_lo = (uint64_t)lane * kSubBits;  // relative to the wave's first subsequence
      const uint64_t sub_hi = sub_lo + kSubBits;
      const uint64_t limit = body_bits - g * 64 * kSubBits;  // end of the stream, same origin
      uint64_t pos = sub_lo + st;
      uint32_t count = 0;
      bool eof = false;
      BitReader br;
      br.init(in, base + pos);
      while (pos < sub_hi && pos < limit) {
        uint32_t len;
        const uint32_t sym = dec_any(L, br.hi(), lut_bits, max_len, len);
        if (sym == 256u) {
          eof = true;
          break;
        }
        br.skip(len);
        pos += len;
        ++count;
      }
      // .crs has no end mark: "eof" then means "this cannot be right" -- a bit pattern that is no code, or a last
      // code that runs past the end of the stream
      if (P.no_eof == 1u && pos > limit) eof = true;
      P.cnt[sub] = count;
      P.eof[sub] = eof ? 1 : 0;
      P.used[sub] = (uint16_t)st;
      if (!eof && pos >= sub_hi && sub + 1 < P.nsub) {
        const uint16_t land = (uint16_t)(pos - sub_hi);
        if (P.start[sub + 1] != land) {
          P.start[sub + 1] = land;
          *P.changed = 1;
        }
      }
      // a PIECE of a stream (mode 2, multi-GPU decode): the last code may run into the next piece's bytes (they are
      // there as look-ahead); where it ends is the next piece's first code boundary
      if (P.no_eof == 2u && sub + 1 == P.nsub) P.start[P.nsub] = eof ? (uint16_t)0xFFFF : (uint16_t)(pos - limit);
    }
  }
}

// first subsequence that holds the end mark (valid once the passes have converged)
__global__ __launch_bounds__(256) void k_sync_eof(SyncParams P) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < P.nsub && P.eof[i]) atomicMin(reinterpret_cast<unsigned long long*>(P.eof_sub), (unsigned long long)i);
}

// symbols per tile of 256 subsequences, nothing counted behind the end mark
__global__ __launch_bounds__(256) void k_sync_tile_sums(SyncParams P) {
  __shared__ unsigned long long ws[4];
  const uint64_t eof_sub = *P.eof_sub;
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  unsigned long long v = (i < P.nsub && i <= eof_sub) ? P.cnt[i] : 0ull;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) P.tile_sum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// absolute bit position of every 64th symbol (the side-car's granularity)
__global__ __launch_bounds__(kDecThreads) void k_sync_index(SyncParams P, uint64_t* __restrict__ seg_abs, uint64_t n_segs) {
  __shared__ DecLds L;
  __shared__ unsigned long long wsum[kDecWaves];
  const int tid = threadIdx.x;
  dec_lds_load(L, P.dt, tid, kDecThreads);
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lut_bits = P.dt->lut_bits, max_len = P.dt->max_len;
  const uint64_t eof_sub = *P.eof_sub;
  const uint64_t sub = (uint64_t)blockIdx.x * 256 + tid;  // one tile of 256 subsequences per workgroup
  const bool valid = sub < P.nsub && sub <= eof_sub;
  const uint32_t c = valid ? P.cnt[sub] : 0u;
  // exclusive prefix of the symbol counts inside the tile
  unsigned long long incl = c;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  unsigned long long first = P.tile_sum[blockIdx.x] + incl - c;  // tile_sum[] holds the exclusive scan by now
  for (int k = 0; k < wave; ++k) first += wsum[k];
  const uint64_t g = (uint64_t)blockIdx.x * kDecWaves + wave;
  uint32_t* in = L.in[wave];
  if (g * 64 >= P.nsub) return;
  const uint64_t base = stage_subs(P, g * 64, in, lane);
  wave_sync();
  if (!valid) return;
  const uint64_t body_bits = P.end_bit - P.body_bit0;
  const uint64_t sub_lo = (uint64_t)lane * kSubBits;
  const uint64_t limit = body_bits - g * 64 * kSubBits;
  uint64_t pos = sub_lo + P.start[sub];
  BitReader br;
  br.init(in, base + pos);
  const uint64_t abs0 = P.body_bit0 + g * 64 * kSubBits;  // stream bit of the wave's first subsequence
  for (uint32_t k = 0; k < c && pos < limit; ++k) {
    const uint64_t sidx = first + k;
    if ((sidx & 63u) == 0 && (sidx >> 6) < n_segs) seg_abs[sidx >> 6] = abs0 + pos;
    uint32_t len;
    (void)dec_any(L, br.hi(), lut_bits, max_len, len);
    br.skip(len);
    pos += len;
  }
}

__global__ __launch_bounds__(256) void k_sync_finalize(const uint64_t* __restrict__ seg_abs, uint64_t n_segs, uint32_t chunk_log2,
                                                       uint64_t* __restrict__ chunk_bit, uint32_t* __restrict__ seg_bit) {
  const uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= n_segs) return;
  const uint64_t c = (s * kSegSymbols) >> chunk_log2;
  const uint64_t s0 = (c << chunk_log2) / kSegSymbols;
  const uint64_t b0 = seg_abs[s0];
  if (s == s0) chunk_bit[c] = b0;
  seg_bit[s] = (uint32_t)(seg_abs[s] - b0);
}

void launch_sync_pass(const SyncParams& p, hipStream_t s) {
  const uint64_t groups = (p.nsub + 63) / 64;
  uint64_t blocks = (groups + kDecWaves - 1) / kDecWaves;
  if (blocks > 256 * 5) blocks = 256 * 5;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(k_sync_pass, dim3((uint32_t)blocks), dim3(kDecThreads), 0, s, p);
}
void launch_sync_counts(const SyncParams& p, uint64_t* d_total, hipStream_t s) {
  const uint32_t tiles = (uint32_t)((p.nsub + 255) / 256);
  hipLaunchKernelGGL(k_sync_eof, dim3(tiles), dim3(256), 0, s, p);
  hipLaunchKernelGGL(k_sync_tile_sums, dim3(tiles), dim3(256), 0, s, p);
  hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, p.tile_sum, tiles, d_total);
}
void launch_sync_index(const SyncParams& p, uint64_t* d_seg_abs, uint64_t n_segs, uint32_t chunk_log2, uint64_t* d_chunk_bit,
                       uint32_t* d_seg_bit, hipStream_t s) {
  const uint32_t tiles = (uint32_t)((p.nsub + 255) / 256);
  hipLaunchKernelGGL(k_sync_index, dim3(tiles), dim3(kDecThreads), 0, s, p, d_seg_abs, n_segs);
  if (n_segs) hipLaunchKernelGGL(k_sync_finalize, dim3((uint32_t)((n_segs + 255) / 256)), dim3(256), 0, s, d_seg_abs, n_segs,
                                 chunk_log2, d_chunk_bit, d_seg_bit);
}

// *dst = (src ? *src : 0) + add   (tiny device-side bookkeeping without a host round trip)
__global__ void k_store_u64(uint64_t* dst, const uint64_t* src, uint64_t add) { *dst = (src ? *src : 0ull) + add; }
void launch_store_u64(uint64_t* d_dst, const uint64_t* d_src_opt, uint64_t add, hipStream_t s) {
  hipLaunchKernelGGL(k_store_u64, dim3(1), dim3(1), 0, s, d_dst, d_src_opt, add);
}

// multi-GPU: this rank's absolute start bit = header bits + body bits of all lower ranks
__global__ void k_shard_start(const ghf_code* code, const uint64_t* totals, int rank, uint64_t* start_bit) {
  uint64_t b = 8ull * (1040ull + 8ull * (uint64_t)code->max_len);
  for (int r = 0; r < rank; ++r) b += totals[r];
  *start_bit = b;
}
void launch_shard_start(const ghf_code* d_code, const uint64_t* d_totals, int rank, uint64_t* d_start_bit, hipStream_t s) {
  hipLaunchKernelGGL(k_shard_start, dim3(1), dim3(1), 0, s, d_code, d_totals, rank, d_start_bit);
}

}  // namespace ghf
