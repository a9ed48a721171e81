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
                                                            uint32_t chunk32, uint32_t nchunks,
                                                            uint32_t* __restrict__ chunk_hist,
                                                            unsigned long long* __restrict__ hist,
                                                            unsigned long long* __restrict__ acc /* [32][256] + done */,
                                                            uint32_t add /* hist += instead of hist = */) {
  static_assert(kHistRep == 32, "replica index is lane % 32");
  __shared__ uint32_t lh[256 * kHistRep];
  __shared__ bool s_last;
  const uint32_t tid = threadIdx.x;
  const uint32_t rep = tid & 31u;
  uint32_t prev = 0;
  unsigned long long total = 0;
  const uint64_t chunk = chunk32;
  // the fast phase's first two vectors are requested BEFORE the counters are cleared: a workgroup's first HBM round trip
  // hides behind the 32 KiB of LDS stores and the barrier instead of following them
  const uint32_t V = chunk32 >> 12;  // vectors per thread per chunk (256 threads x 16 B = 4 KiB); any count >= 4
  const uint32_t nfullchunks = (uint32_t)(n / chunk);
  const bool fast = V >= 4u && (chunk32 & 4095u) == 0 && (((uintptr_t)in) & 15u) == 0 && nfullchunks > 0;
  auto vptr = [&](uint32_t c, uint32_t j) -> const uint4* {
    if (c >= nfullchunks) c = nfullchunks - 1;  // past the end: redundant, harmless loads
    return reinterpret_cast<const uint4*>(in + (uint64_t)c * chunk) + (uint64_t)j * kHistThreads + tid;
  };
  uint4 A = make_uint4(0, 0, 0, 0), B = A;
  if (fast) {
    // non-temporal loads: the input streams past the Infinity Cache instead of through it (4 GiB: 0.79 -> 0.69 ms, i.e.
    // 6.2 TB/s of reads; nothing changes at 256 MiB, where the pipelined neighbours' write-backs set the pace)
    A = load_stream(vptr(blockIdx.x, 0));
    B = load_stream(vptr(blockIdx.x, 1));
  }
  for (uint32_t i = tid; i < 256 * kHistRep; i += kHistThreads) lh[i] = 0;
  __syncthreads();

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
  if (fast) {
    // (a class's workgroups sit on all eight XCDs -- consecutive workgroups go to consecutive XCDs -- so that no XCD is
    //  tied to a fixed share of the chunks; see k_decode)
    const uint32_t ncls = gridDim.x < 16u ? 1u : (gridDim.x < 128u ? gridDim.x >> 3 : 16u);
    const uint32_t cls = (blockIdx.x >> 3) % ncls;
    unsigned long long* tick = acc + 32 * 256 + 16 + cls * 16;
    // the first two chunks of a workgroup are fixed, so its first loads go out before any counter has answered;
    // tickets number the chunks behind those 2 * gridDim.x
    auto draw = [&]() -> uint32_t { return 2u * gridDim.x + (uint32_t)atomicAdd(tick, 1ull) * ncls + cls; };  // thread 0 only
    uint32_t cur = blockIdx.x;
    uint32_t nxt = blockIdx.x + gridDim.x;
    // vector g of the thread's STREAM: the chunk's own vectors, then the next chunk's
    auto vat = [&](uint32_t g) -> const uint4* { return g < V ? vptr(cur, g) : vptr(nxt, g - V); };
    const uint32_t Vmain = V & ~3u, Vrest = V & 3u;  // (wave-uniform, the same for every chunk of the launch)
    while (cur < nfullchunks) {
      uint32_t t_next = 0;
      if (tid == 0) t_next = draw();  // the chunk after next; the atomic returns long before it is needed
      for (uint32_t j = 0; j < Vmain; j += 4) {
        const uint4 C = load_stream(vat(j + 2)), D = load_stream(vat(j + 3));
        hist_vec(lh, rep, A);
        hist_vec(lh, rep, B);
        A = load_stream(vat(j + 4));
        B = load_stream(vat(j + 5));
        hist_vec(lh, rep, C);
        hist_vec(lh, rep, D);
      }
      // A, B = vectors Vmain, Vmain + 1 of the stream: the chunk's last one to three vectors, then the next chunk's first two
      if (Vrest == 1u) {
        hist_vec(lh, rep, A);
        A = B;  // (was the next chunk's vector 0)
        B = load_stream(vptr(nxt, 1));
      } else if (Vrest == 2u) {
        hist_vec(lh, rep, A);
        hist_vec(lh, rep, B);
        A = load_stream(vptr(nxt, 0));
        B = load_stream(vptr(nxt, 1));
      } else if (Vrest == 3u) {
        const uint4 C = load_stream(vptr(cur, V - 1));
        hist_vec(lh, rep, A);
        hist_vec(lh, rep, B);
        A = load_stream(vptr(nxt, 0));
        B = load_stream(vptr(nxt, 1));
        hist_vec(lh, rep, C);
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
    const uint64_t base = (uint64_t)c * chunk;
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
    // all 32 replica loads in flight at once (one L2 round trip at the very end of the launch, not four), then the stores
    unsigned long long part[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) part[r] = __hip_atomic_load(&acc[r * 256 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long sum = 0;
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      sum += part[r];
      __hip_atomic_store(&acc[r * 256 + tid], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    hist[tid] = add ? hist[tid] + sum : sum;
    if (tid < 16) __hip_atomic_store(&acc[32 * 256 + 16 + tid * 16], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) {
      hist[256] = 1;  // include/encoder.h:128 end-of-stream mark counts once
      __hip_atomic_store(&acc[32 * 256], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

void launch_histogram(const uint8_t* d_in, uint64_t n, uint32_t chunk, uint32_t nchunks, uint32_t* d_chunk_hist,
                      uint64_t* d_hist, uint64_t* d_acc, bool add, hipStream_t s) {
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
  hipLaunchKernelGGL(k_histogram, dim3(grid), dim3(kHistThreads), 0, s, d_in, n, chunk, nchunks, d_chunk_hist,
                     reinterpret_cast<unsigned long long*>(d_hist), reinterpret_cast<unsigned long long*>(d_acc),
                     add ? 1u : 0u);
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
// sibling comparisons alone (the value being re-inserted plays no part until the final __push_heap).  With
// 1-based node numbers t (slot[t]; children 2t and 2t+1 sit on one 16-byte boundary), lane l OWNS the
// children of nodes l and 64 + l (lane 0: nodes 128 and 64) -- 128 parents cover a heap of 257 entries:
//   1. ONE LDS round trip: every lane loads the children pairs of its two nodes; root and last entry are
//      read along.  Two ballots = a 128-bit map "preferred child is the right one" for the whole heap;
//   2. the scalar unit follows the map from the root down: t = 2t + bit[t], two or three scalar
//      instructions a level, no memory;
//   3. everything the pop has to move is ALREADY in registers: the entry that moves up into path node u is
//      u's preferred child, held by u's owner.  Every lane checks whether its nodes are on the path
//      (t_leaf >> shift == u), one ballot against the re-inserted value tells where __push_heap stops, and
//      the owners store: slot[u] = child entry above the stop, slot[stop] = value.
// Identical result to the sequential code above in a third of its time; heap_pop/heap_sift_up stay as the
// executable specification.
__device__ __forceinline__ u64t wave_uniform(u64t x) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32));
  return ((u64t)hi << 32) | lo;
}

__device__ __forceinline__ u64t lane_above(u64t x) {  // value held by lane + 1 (same row of 16 lanes)
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)x, 0x101, 0xF, 0xF, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(x >> 32), 0x101, 0xF, 0xF, false);
  return ((u64t)hi << 32) | lo;
}

__device__ __forceinline__ u64t wave_heap_pop(HeapLds& h, int& n, int lane) {
  const int len = __builtin_amdgcn_readfirstlane(n) - 1;  // entries left behind (wave-uniform: keeps the walk on the scalar unit)
  n = len;
  const uint32_t tA = lane ? (uint32_t)lane : 128u, tB = 64u + (uint32_t)lane;
  // (all four reads go out together and are waited for once: a pop is two LDS round trips, this one and the path's below)
  const U64x2 ca = *reinterpret_cast<const U64x2*>(&h.slot[2 * tA]);
  const U64x2 cb = *reinterpret_cast<const U64x2*>(&h.slot[2 * tB]);
  const u64t top_v = h.slot[1];
  const u64t value_v = h.slot[len + 1];  // (len == 0: slot[1] again, unused)
  __builtin_amdgcn_sched_barrier(0);
  const u64t top = wave_uniform(top_v);
  if (len < 1) return top;
  const u64t value = wave_uniform(value_v);  // the old last entry, re-inserted from the hole
  // 1. "go right" = !comp(right, left) (libstdc++ takes the LEFT child when freq[right] > freq[left])
  const unsigned long long R0 = __ballot(!heap_gt(ca.y, ca.x));  // bit l = node l (l >= 1), bit 0 = node 128
  const unsigned long long R1 = __ballot(!heap_gt(cb.y, cb.x));  // bit l = node 64 + l
  // 2. walk.  0-based: while (hole < (len - 1) / 2) -> child; 1-based t = hole + 1: while (t <= lim).  With D = depth of
  //    the last position, every node above depth D-1 has two children: the first D-1 steps need no bounds test.
  const uint32_t lim = (uint32_t)(len - 1) >> 1;
  const int D = 31 - __clz(len);
  // seven levels unconditionally (two scalar instructions each), then cut the path back to its first D-1 steps: the
  // prefix of a longer walk IS the shorter walk (what lies below may be stale slots; it is shifted out)
  // (t = 2 t + bit t of the map: s_bitcmp1_b64 puts the bit into SCC -- it looks at the low six bits of t only, which is
  //  what level 7 wants -- and s_addc_u32 t, t, t adds it in: 14 scalar instructions for the 7 levels; the compiler's
  //  shift / and / shift / or rendering of the same took 40)
  uint32_t t = 1;
  asm("s_bitcmp1_b64 %1, %0\n\ts_addc_u32 %0, %0, %0\n\t"
      "s_bitcmp1_b64 %1, %0\n\ts_addc_u32 %0, %0, %0\n\t"
      "s_bitcmp1_b64 %1, %0\n\ts_addc_u32 %0, %0, %0\n\t"
      "s_bitcmp1_b64 %1, %0\n\ts_addc_u32 %0, %0, %0\n\t"
      "s_bitcmp1_b64 %1, %0\n\ts_addc_u32 %0, %0, %0\n\t"
      "s_bitcmp1_b64 %1, %0\n\ts_addc_u32 %0, %0, %0\n\t"  // nodes 1..63
      "s_bitcmp1_b64 %2, %0\n\ts_addc_u32 %0, %0, %0"        // nodes 64..127
      : "+s"(t)
      : "s"(R0), "s"(R1)
      : "scc");
  const int steps = D - 1 > 0 ? D - 1 : 0;
  t >>= 7 - steps;
  if (t <= lim) t = 2u * t + (uint32_t)(((t < 64u || t == 128u ? R0 : R1) >> (t & 63u)) & 1ull);  // depth D-1 where both children exist
  t <<= (((uint32_t)len & 1u) ^ 1u) & (uint32_t)(t == ((uint32_t)len >> 1));  // lone left child of an even-length heap (kept in scalar arithmetic)
  const int k = 31 - __clz(t);  // depth of the hole
  // 3. lane j < k takes path node u_j = t >> (k - j): ONE more LDS read (the children pair of u_j) gives cv_j, the entry that
  //    __adjust_heap moves up into u_j (its preferred child, which is path node u_{j+1}).  __push_heap from the hole then moves
  //    entries back down while comp(entry, value); it stops below the DEEPEST path entry with !comp: with m = that depth + 1
  //    (0: none), u_i receives cv_i for i < m, u_m receives `value`, everything deeper keeps what it had.
  const uint32_t sh = (uint32_t)(k - lane) & 31u;
  const bool onp = lane < k;
  const uint32_t u = onp ? (t >> sh) : 1u;
  const U64x2 c = *reinterpret_cast<const U64x2*>(&h.slot[2 * u]);
  const u64t cv = ((t >> ((sh - 1u) & 31u)) & 1u) ? c.y : c.x;
  const unsigned long long S = __ballot(!heap_gt(cv, value)) & ((1ull << k) - 1ull);  // lanes 0..k-1 are on the path
  const int m = S ? 64 - __clzll((long long)S) : 0;
  if (lane <= m) h.slot[lane < m ? u : (t >> ((uint32_t)(k - m) & 31u))] = lane < m ? cv : value;
  wave_sync();  // the next heap operation reads, in OTHER lanes, what these lanes stored (without the fence the compiler may
                // forward a lane's own store to its next load and let the other lanes' load overtake the store)
  return top;
}

// priority_queue::push(e) onto a heap of n entries: __push_heap from position n.  In two steps, so that a caller can put
// other LDS reads into the same round trip: lane j = 1..depth reads the j-th ancestor of the new position ...
struct PushLoad {
  u64t pe;
  int aj, depth;
  bool on;
};
__device__ __forceinline__ PushLoad wave_heap_push_load(const HeapLds& h, int n_, int lane) {
  const int n = __builtin_amdgcn_readfirstlane(n_);
  PushLoad L;
  L.depth = 31 - __clz(n + 1);  // number of ancestors of position n
  L.on = lane >= 1 && lane <= L.depth;
  L.aj = ((n + 1) >> lane) - 1;  // lane 0: n itself
  L.pe = L.on ? h.slot[L.aj + 1] : 0ull;
  return L;
}
// ... and the entries above the first ancestor that stays move down one level each (one ballot, one DPP shift)
__device__ __forceinline__ void wave_heap_push_finish(HeapLds& h, const PushLoad& L, u64t e, int lane) {
  const bool stop = L.on && !heap_gt(L.pe, e);
  const unsigned long long sm = __ballot(stop);
  const int t = sm ? (__ffsll((long long)sm) - 1) - 1 : L.depth;  // entries of lanes 1..t move down one level
  const u64t up = lane_above(L.pe);
  if (lane <= t) h.slot[L.aj + 1] = (lane < t) ? up : e;
  wave_sync();
}
__device__ __forceinline__ void wave_heap_push(HeapLds& h, int n_, u64t e, int lane) {
  const PushLoad L = wave_heap_push_load(h, n_, lane);
  wave_heap_push_finish(h, L, e, lane);
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
                                                   int* __restrict__ status, uint32_t empty_ok) {
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
      // one LDS round trip for the push's ancestors and the two groups' current tree nodes (every lane reads the same
      // two words: no divergent block, no wait of its own)
      const PushLoad pl = wave_heap_push_load(heap, n, lane);
      const uint16_t g1 = heap.cur[s1], g2 = heap.cur[s2];
      if (lane == 0) {
        heap.parent[g1] = (uint16_t)node;  // .cc:316-329: both groups one level deeper ...
        heap.parent[g2] = (uint16_t)node;
        heap.cur[s2] = (uint16_t)node;     // ... and merged under the second popped index
      }
      const u64t f = (e1 >> 9) + (e2 >> 9);                              // .cc:331
      wave_heap_push_finish(heap, pl, (f << 9) | (u64t)s2, lane);        // .cc:333
      ++n;
    }
  }
  __syncthreads();
  if (s_ndata == 0 && !empty_ok) {  // empty input: undefined in the reference
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
  // GHF_EMPTY_OK: the lone end mark (no merge happened, its depth is 0) gets the one-bit code "0" -- our definition
  if (s_ndata == 0 && lane == 0) len[4] = 1;
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
    hipLaunchKernelGGL(k_build_code<true>, dim3(1), dim3(64), 0, s, reinterpret_cast<const unsigned long long*>(d_hist), d_code, d_status,
                       flags & GHF_EMPTY_OK);
  else
    hipLaunchKernelGGL(k_build_code<false>, dim3(1), dim3(64), 0, s, reinterpret_cast<const unsigned long long*>(d_hist), d_code, d_status,
                       flags & GHF_EMPTY_OK);
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
  uint32_t code_hi[256];                 // bits 32..63 of a leaf's code
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
      if (len <= 64) {
        out_code->length[v] = len;
        out_code->codeword[v] = (uint32_t)code;
        T.code_hi[v] = (uint32_t)(code >> 32);
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
  __syncthreads();
  // symbols that do not occur, the end-mark slot and the canonical-only tables of ghf_code.  A tree deeper than 32
  // (huff_tree.cc:157-170 keeps codes as strings of any length) puts bits 32..63 of every code into symbol[], which this
  // format has no other use for: K5's long-code kernel reads them there.
  const bool deep = T.max_len > 32u;
  for (int s = lane; s < GHF_NSYM; s += 64) {
    const bool absent = s == 256 || heap.parent[s] == 0;
    if (absent) {
      out_code->length[s] = 0;
      out_code->codeword[s] = 0;
    }
    out_code->symbol[s] = deep ? (absent ? 0u : T.code_hi[s]) : 0xFFFFFFFFu;
  }
  for (int i = lane; i < 64; i += 64) {
    out_code->first_code[i] = 0;
    out_code->start_pos[i] = 0;
  }
  if (T.max_len > 64u) {  // (needs more than 2^44 input bytes: Fibonacci counts)
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

__global__ __launch_bounds__(256) void k_chunk_bits_direct(const uint8_t* __restrict__ in, uint64_t n, uint32_t chunk32,
                                                           uint32_t nchunks, const ghf_code* __restrict__ code,
                                                           uint64_t* __restrict__ bits) {
  __shared__ uint32_t ll[256];
  ll[threadIdx.x] = code->length[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint64_t chunk = chunk32;
  for (uint32_t c = wave; c < nchunks; c += nwaves) {
    const uint64_t base = (uint64_t)c * chunk;
    const uint64_t len = (n - base < chunk) ? (n - base) : chunk;
    unsigned long long b = 0;
    // 16 bytes per lane and load where the chunk is 16-byte aligned (every piece of a file, every shard: this is the K4 of
    // the callers that have no per-chunk counts from K1), single bytes for a ragged head and tail
    const uint8_t* p = in + base;
    uint64_t head = (16u - (uint32_t)((uintptr_t)p & 15u)) & 15u;
    if (head > len) head = len;
    if ((uint64_t)lane < head) b += ll[p[lane]];
    const uint4* pv = reinterpret_cast<const uint4*>(p + head);
    const uint64_t nvec = (len - head) >> 4;
    uint32_t acc = 0;
    for (uint64_t i = lane; i < nvec; i += 64) {
      const uint4 v = pv[i];  // (K5 reads the same bytes next: let them stay in the caches)
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        acc += ll[w[k] & 0xFFu] + ll[(w[k] >> 8) & 0xFFu] + ll[(w[k] >> 16) & 0xFFu] + ll[w[k] >> 24];
      if ((i >> 6) % 4096u == 4095u) {  // (64 lanes x 16 symbols x 255 bits per round: far from 2^32 in 4096 rounds)
        b += acc;
        acc = 0;
      }
    }
    b += acc;
    const uint64_t tail0 = head + (nvec << 4);
    if (tail0 + (uint64_t)lane < len) b += ll[p[tail0 + lane]];
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

void launch_scan(uint64_t* d_v, uint32_t count, uint64_t* d_total, hipStream_t s) {
  hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, d_v, count, d_total);
}

void launch_plan(const uint8_t* d_in, uint64_t n, uint32_t chunk, uint32_t nchunks, const uint32_t* d_chunk_hist,
                 const ghf_code* d_code, uint64_t* d_chunk_off, uint64_t* d_total_bits, hipStream_t s) {
  uint32_t blocks = (nchunks + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  if (blocks == 0) blocks = 1;
  if (d_chunk_hist)
    hipLaunchKernelGGL(k_chunk_bits, dim3(blocks), dim3(256), 0, s, d_chunk_hist, nchunks, d_code, d_chunk_off);
  else
    hipLaunchKernelGGL(k_chunk_bits_direct, dim3(blocks), dim3(256), 0, s, d_in, n, chunk, nchunks, d_code,
                       d_chunk_off);
  hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, d_chunk_off, nchunks, d_total_bits);
}

// *dst = (src ? *src : 0) + add   (tiny device-side bookkeeping without a host round trip)
__global__ void k_store_u64(uint64_t* dst, const uint64_t* src, uint64_t add) { *dst = (src ? *src : 0ull) + add; }
void launch_store_u64(uint64_t* d_dst, const uint64_t* d_src_opt, uint64_t add, hipStream_t s) {
  hipLaunchKernelGGL(k_store_u64, dim3(1), dim3(1), 0, s, d_dst, d_src_opt, add);
}

// *dst = *src for a 16-bit word (K6: the landing bit of a piece, fetched with the counts in one host round trip)
__global__ void k_load_u16(uint64_t* dst, const uint16_t* src) { *dst = *src; }
void launch_load_u16(uint64_t* d_dst, const uint16_t* d_src, hipStream_t s) { hipLaunchKernelGGL(k_load_u16, dim3(1), dim3(1), 0, s, d_dst, d_src); }

// multi-GPU: this rank's absolute start bit = header bits + body bits of all lower ranks
__global__ void k_shard_start(const ghf_code* code, const uint64_t* totals, int rank, uint64_t* start_bit) {
  uint64_t b = 8ull * (1040ull + 8ull * (uint64_t)code->max_len);
  for (int r = 0; r < rank; ++r) b += totals[r];
  *start_bit = b;
}
void launch_shard_start(const ghf_code* d_code, const uint64_t* d_totals, int rank, uint64_t* d_start_bit, hipStream_t s) {
  hipLaunchKernelGGL(k_shard_start, dim3(1), dim3(1), 0, s, d_code, d_totals, rank, d_start_bit);
}

// ------------------------------------------------------------------------------------------------
// Bandwidth probe (ghf_copy_d2d): what a kernel that only MOVES bytes reaches with this path's own access shape --
// 16 bytes per lane, four loads in flight per lane, every workgroup streaming through a contiguous slab of its own (what
// K5's and K7's waves do with their chunks and groups), optional non-temporal hints.  Of the shapes scratch/membench.hip
// tries this is the fastest on MI355X (4 GiB: 5.3-5.5 TB/s of read + write; grid-stride with the same loads 4.6-4.9;
// profiles/r03/membench.txt).  bench.py prices the codec kernels against it (and against the 8 TB/s of the data sheet).
// ------------------------------------------------------------------------------------------------
template <bool NT>
__global__ __launch_bounds__(256) void k_stream_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t nvec) {
  const uint64_t per = (nvec + gridDim.x - 1) / gridDim.x;
  const uint64_t lo = (uint64_t)blockIdx.x * per;
  const uint64_t end = lo + per < nvec ? lo + per : nvec;
  uint64_t i = lo + threadIdx.x;
  auto ld = [&](uint64_t k) -> uint4 { return NT ? load_stream(src + k) : src[k]; };
  auto st = [&](uint64_t k, const uint4& v) {
    if (NT) store_stream(dst + k, v);
    else dst[k] = v;
  };
  for (; i + 3 * 256 < end; i += 4 * 256) {
    const uint4 a = ld(i), b = ld(i + 256), c = ld(i + 512), d = ld(i + 768);
    st(i, a); st(i + 256, b); st(i + 512, c); st(i + 768, d);
  }
  for (; i < end; i += 256) st(i, ld(i));
}
__global__ void k_tail_copy(const uint8_t* src, uint8_t* dst, uint64_t from, uint64_t n) {
  const uint64_t i = from + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
void launch_stream_copy(const uint8_t* d_src, uint8_t* d_dst, uint64_t n, bool nt, hipStream_t s) {
  const uint64_t nvec = n / 16;
  if (nvec) {
    uint64_t blocks = (nvec + 255) / 256;
    if (blocks > 4096) blocks = 4096;  // slabs: 16 workgroups of 4 waves per CU
    if (nt) hipLaunchKernelGGL(k_stream_copy<true>, dim3((uint32_t)blocks), dim3(256), 0, s, reinterpret_cast<const uint4*>(d_src), reinterpret_cast<uint4*>(d_dst), nvec);
    else hipLaunchKernelGGL(k_stream_copy<false>, dim3((uint32_t)blocks), dim3(256), 0, s, reinterpret_cast<const uint4*>(d_src), reinterpret_cast<uint4*>(d_dst), nvec);
  }
  if (n & 15) hipLaunchKernelGGL(k_tail_copy, dim3(1), dim3(16), 0, s, d_src, d_dst, nvec * 16, n);
}

}  // namespace ghf
