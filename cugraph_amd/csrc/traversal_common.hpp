// Frontier expansion shared by the single-GPU drivers (traversal.hip) and the partitioned engine (traversal_mg.hip):
// one wave-cooperative, edge-balanced walk over the out-edges of a vertex list, parameterised by a per-edge functor.
// Replaces extract_transform_if_v_frontier_e (cpp/include/cugraph/prims/detail/extract_transform_if_v_frontier_e.cuh:127/309/422).
#pragma once
#include "common.hpp"

namespace cga {
namespace {

constexpr int TV_BLOCK = 256;
constexpr int TV_WAVES = TV_BLOCK / 64;
constexpr int32_t BIG_DEG = 2048;      // rows at least this long are deferred to k_*_big when the frontier is wide ...
constexpr int32_t BIG_DEG_NARROW = 64; // ... and rows at least THIS long when it is narrow: a wavefront walks its "whole-wave" rows one after
                                       // the other, so a frontier of a few hundred vertices with 10^2..10^3 out-edges each (a level or a
                                       // relaxation round right after the source) ran on a handful of wavefronts: 3.6 ms for one BFS level
                                       // of 1141 vertices at RMAT-24, 0.5 ms for one of 30.  Deferred rows get one workgroup per 4096 edges.
constexpr int32_t BIG_SEG = 4096;  // edges per deferred (row, segment) work unit
// ... of a NARROW frontier (round 5, BFS): one frontier vertex with 10^5 out-edges is 25 work units of 4096 edges -- 25 workgroups of 256
// threads walking 16 dependent probes each while 2 000 workgroup slots idle.  Measured at RMAT-24 (32 roots, same session,
// profiles/r5p_bfs_seg.txt): 4096 -> 1.347 / 1.346 ms, 1024 -> 1.324, 512 -> 1.339 / 1.334, 256 -> 1.359: a per cent; 1024 it is.
constexpr int32_t BIG_SEG_NARROW = 1024;
__host__ __device__ inline int32_t big_seg_for_deg(int32_t big_deg, int32_t narrow_seg) { return big_deg == 64 ? narrow_seg : 4096; }

// Plain sums (edges inspected, degree sums of the discoveries, discoveries of the bottom-up kernel) are added once per wavefront
// at kernel exit.  With one counter per quantity that is 10^4..10^5 atomics on ONE 64-byte line per launch, and same-line atomics
// retire one after the other in their L2 channel (~12 ns each): a BFS bottom-up level of 16 Ki wavefronts took 0.85 ms whatever
// it did, 1.6 ms with twice the wavefronts.  The sums therefore go to one of CNT_REPLICAS lines picked by wavefront; the host
// folds them after the read-back (counters_t::fold).  Cursors (n_next / n_far / n_big of the queue-producing kernels) cannot be
// replicated -- they hand out positions -- but they are touched once per 64+ items, not once per wavefront.
constexpr int CNT_REPLICAS = 32;
struct counter_sums_t {  // one 64-byte line
  unsigned long long edges, out_edges, in_edges, n_found;
  unsigned long long pad[4];
};
struct counters_t {  // device-resident, zeroed per step
  // the three queue cursors sit in three different 64-byte lines: appends to the near and the far queue of an SSSP round would
  // otherwise queue up behind each other in one L2 channel
  uint32_t n_next;   // size of the next (near) frontier
  uint32_t padl0[15];
  uint32_t n_far;    // size of the far pile (SSSP)
  uint32_t padl1[15];
  uint32_t n_set;    // SSSP (light / heavy): vertices that entered the current bucket (their heavy edges are relaxed when it closes)
  uint32_t padl2[15];
  uint32_t n_big;    // deferred high-degree vertices
  uint32_t pad;
  unsigned long long edges;  // edges inspected
  uint32_t far_min_bits_lo;  // (SSSP split) min distance bits kept in far (32-bit types)
  uint32_t pad2;
  unsigned long long far_min_bits64;
  unsigned long long out_edges;  // BFS: sum of out-degrees of the vertices discovered in this level (top-down cost of the next)
  unsigned long long in_edges;   // BFS: sum of their in-degrees (they leave the bottom-up work)
  unsigned long long pad3[2];
  counter_sums_t rep[CNT_REPLICAS];
  __host__ __device__ void fold()  // host, after the read-back: replicas -> the plain fields
  {
    for (int r = 0; r < CNT_REPLICAS; ++r) {
      edges += rep[r].edges; out_edges += rep[r].out_edges; in_edges += rep[r].in_edges; n_next += (uint32_t)rep[r].n_found;
      rep[r] = counter_sums_t{};
    }
  }
};
static_assert(sizeof(counters_t) == 4 * 64 + CNT_REPLICAS * 64 && sizeof(counters_t) <= 3072, "counters_t: one line + the replica lines; read back through the pinned page");

// the calling wavefront's replica line (all lanes get the same one)
__device__ __forceinline__ counter_sums_t* cnt_replica(counters_t* cnt)
{
  unsigned const w = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
  return &cnt->rep[w % CNT_REPLICAS];
}

__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, int lane, uint32_t* total)
{
  uint32_t inc = v;
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(inc, o);
    if (lane >= o) inc += t;
  }
  *total = __shfl(inc, 63);
  return inc - v;
}

// wave-aggregated append of `flag` lanes' values to a queue
__device__ __forceinline__ void wave_push(bool flag, int32_t value, int32_t* q, uint32_t* counter, int lane)
{
  uint64_t m = __ballot(flag);
  if (m == 0) return;
  uint32_t base = 0;
  int leader    = __ffsll((unsigned long long)m) - 1;
  if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
  base = __shfl(base, leader);
  if (flag) q[base + __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))))] = value;
}

// Appends staged per wavefront in LDS: one global atomicAdd (and one coalesced copy) per WQ_CAP entries instead of one per
// wavefront and call.  In a wide frontier the per-call atomics all hit the SAME counter word and serialise in one L2 channel
// (profiles: k_sssp_expand 17.8 ms for one relaxation round at RMAT-24 with per-call appends).  No wave-uniform register
// state: lanes that sit out a call (ragged loop tails) simply do not take part, the fill counter lives in LDS.
#ifndef CGA_WQ_CAP
#define CGA_WQ_CAP 256
#endif
constexpr int WQ_CAP = CGA_WQ_CAP;  // (1024 measured equal for BFS, 5 % slower for SSSP: the cursor is not what a round waits for)
template <int NQ>
struct wave_queue_storage {
  int32_t buf[NQ][TV_WAVES][WQ_CAP];
  uint32_t fill[NQ][TV_WAVES];
  __device__ __forceinline__ void init()  // whole workgroup, before first use
  {
    if (threadIdx.x < NQ * TV_WAVES) (&fill[0][0])[threadIdx.x] = 0;
    __syncthreads();
  }
};
struct wave_queue {
  int32_t* buf{nullptr};    // LDS: WQ_CAP entries owned by this wavefront
  uint32_t* fill{nullptr};  // LDS: its fill counter
  int32_t* q{nullptr};      // global queue
  uint32_t* counter{nullptr};
  template <int NQ>
  __device__ __forceinline__ wave_queue(wave_queue_storage<NQ>& st, int k, int32_t* q_, uint32_t* counter_)
    : buf(st.buf[k][threadIdx.x >> 6]), fill(&st.fill[k][threadIdx.x >> 6]), q(q_), counter(counter_)
  {
  }
  __device__ __forceinline__ void push(bool flag, int32_t value)
  {
    uint64_t const m = __ballot(flag);
    if (m == 0) return;
    int const lane      = threadIdx.x & 63;
    uint32_t const c    = (uint32_t)__popcll(m);
    int const leader    = __ffsll((unsigned long long)m) - 1;
    uint32_t const rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    uint32_t base       = 0;
    if (lane == leader) base = atomicAdd(fill, c);  // LDS
    base = __shfl(base, leader);
    if (base + c <= (uint32_t)WQ_CAP) {
      if (flag) buf[base + rank] = value;
      return;
    }
    // full: the staged entries and the new ones go out together, copied by the lanes that are active here
    uint32_t g = 0;
    if (lane == leader) { g = atomicAdd(counter, base + c); *fill = 0; }
    g = __shfl(g, leader);
    uint64_t const act = __ballot(true);
    uint32_t const na = (uint32_t)__popcll(act), ar = (uint32_t)__popcll(act & ((1ull << lane) - 1ull));
    for (uint32_t i = ar; i < base; i += na) q[g + i] = buf[i];
    if (flag) q[g + base + rank] = value;
  }
  __device__ __forceinline__ void flush()  // every lane of the wavefront, at the end of the kernel
  {
    __builtin_amdgcn_wave_barrier();
    uint32_t const n = *fill;
    if (n == 0) return;
    int const lane = threadIdx.x & 63;
    uint32_t g = 0;
    if (lane == 0) g = atomicAdd(counter, n);
    g = __shfl(g, 0);
    for (uint32_t i = lane; i < n; i += 64) q[g + i] = buf[i];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) *fill = 0;
  }
};

// Edge positions are UNSIGNED 32-bit (eoff_t): the offsets arrays hold values up to 2^32 - 1 in their int32 words (a graph of
// 2^31 .. 2^32 - 1 edges -- symmetrised RMAT-26 -- is addressable; a row has fewer than 2^31 edges).
using eoff_t = uint32_t;
__device__ __forceinline__ eoff_t eoff(int32_t const* offsets, int64_t v) { return (eoff_t)offsets[v]; }

// Expands the frontier `q[0..n)` (q == nullptr: vertices 0..n-1): calls f(u, v, edge_position) for every
// out-edge of every frontier vertex for which keep(u) is true.  Vertices of degree >= BIG_DEG are pushed to
// bigq for k_expand_big.
template <typename Keep, typename F>
__device__ __forceinline__ void expand_frontier(int32_t const* q, int64_t n, int32_t const* offsets, int32_t const* indices,
                                                int32_t* bigq, counters_t* cnt, Keep keep, F& f, int32_t big_deg = BIG_DEG,
                                                int32_t const* row_end = nullptr,  // row u = [offsets[u], row_end ? row_end[u] : offsets[u + 1])
                                                int32_t seg = BIG_SEG)             // edges per deferred work unit (the same value goes to expand_big)
{
  __shared__ uint32_t s_scan[TV_WAVES][64];
  __shared__ eoff_t s_beg[TV_WAVES][64];
  __shared__ int32_t s_u[TV_WAVES][64];
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t const gwave  = (int64_t)blockIdx.x * TV_WAVES + wave;
  int64_t const nwaves = (int64_t)gridDim.x * TV_WAVES;
  unsigned long long inspected = 0;
  for (int64_t base = gwave * 64; base < n; base += nwaves * 64) {
    int64_t i = base + lane;
    int32_t u = -1, deg = 0;
    eoff_t beg = 0;
    if (i < n) {
      u = q ? q[i] : (int32_t)i;
      if (keep(u)) { beg = eoff(offsets, u); deg = (int32_t)((row_end ? eoff(row_end, u) : eoff(offsets, u + 1)) - beg); } else { u = -1; }
    }
    // deferred: huge rows, cut into BIG_SEG-edge segments (one workgroup of k_*_big each)
    bool big = deg >= big_deg;
    if (big) {
      uint32_t nseg = ((uint32_t)deg + (uint32_t)seg - 1) / (uint32_t)seg;
      uint32_t at   = atomicAdd(&cnt->n_big, nseg);
      for (uint32_t sgm = 0; sgm < nseg; ++sgm) { bigq[2 * (at + sgm)] = u; bigq[2 * (at + sgm) + 1] = (int32_t)sgm; }
    }
    // whole-wave rows
    uint64_t mid = __ballot(deg >= 64 && !big);
    while (mid) {
      int src     = __ffsll((unsigned long long)mid) - 1;
      mid &= mid - 1;
      int32_t uu = __shfl(u, src), d = __shfl(deg, src);
      eoff_t const b = (eoff_t)__shfl((int)beg, src);
      for (int32_t p = lane; p < d; p += 64) f(uu, indices[b + (eoff_t)p], b + (eoff_t)p);
      inspected += (lane == 0) ? (unsigned long long)d : 0ull;
    }
    // flattened small rows
    uint32_t sd = (deg < 64) ? (uint32_t)deg : 0u, total;
    uint32_t ex = wave_excl_scan(sd, lane, &total);
    s_scan[wave][lane] = ex;
    s_beg[wave][lane]  = beg;
    s_u[wave][lane]    = u;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t t = lane; t < total; t += 64) {
      // owner = last lane j with s_scan[j] <= t (rows of degree 0 share a scan value with their successor;
      // the search returns the last of them, whose degree is > 0 by construction of `total`)
      int lo = 0, hi = 63;
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        int m = (lo + hi + 1) >> 1;
        if (s_scan[wave][m] <= t) lo = m; else hi = m - 1;
      }
      eoff_t const p = s_beg[wave][lo] + (t - s_scan[wave][lo]);
      f(s_u[wave][lo], indices[p], p);
    }
    __builtin_amdgcn_wave_barrier();
    inspected += (lane == 0) ? (unsigned long long)total : 0ull;
  }
  if (lane == 0 && inspected) atomicAdd(&cnt_replica(cnt)->edges, inspected);
}

template <typename F>
__device__ __forceinline__ void expand_big(int32_t const* bigq, int32_t const* offsets, int32_t const* indices, counters_t* cnt, F& f,
                                           int32_t const* row_end = nullptr, int32_t seg = BIG_SEG)
{  // one workgroup per (row, segment) pair: rows of 10^3..10^6 edges all get parallelism proportional to their length
  uint32_t const nseg = cnt->n_big;
  unsigned long long inspected = 0;
  for (uint32_t k = blockIdx.x; k < nseg; k += gridDim.x) {
    int32_t const u = bigq[2 * k], sgm = bigq[2 * k + 1];
    eoff_t const row_b = eoff(offsets, u), row_e = row_end ? eoff(row_end, u) : eoff(offsets, u + 1);
    eoff_t const b = row_b + (eoff_t)sgm * (eoff_t)seg;                 // (b <= row_e: the segment exists)
    eoff_t const len = min(row_e - b, (eoff_t)seg);
    for (eoff_t p = threadIdx.x; p < len; p += blockDim.x) f(u, indices[b + p], b + p);
    if (threadIdx.x == 0) inspected += (unsigned long long)len;
  }
  if (threadIdx.x == 0 && inspected) atomicAdd(&cnt_replica(cnt)->edges, inspected);
}


// ---- the same two walks with EX_U edges in flight per lane.  A relaxation is a chain of dependent memory round trips (neighbour
// id -> its distance / visited word -> atomic): with one edge per lane and step a wavefront has 64 requests outstanding and spends
// the round trip waiting (measured: adding ONE more cached load to the chain made an SSSP round 15 % slower; removing the returned
// atomic did not help -- latency, not bytes or atomics, bounded the wide rounds).  The functor is therefore split in two:
//   cand_t pre(u, v, p)          loads and tests only, BRANCH-FREE (v < 0: no edge; u, v, p are then clamped to loadable positions by
//                                the caller / the functor): the loads of the EX_U pre() calls of a step are issued back to back
//   tok_t  mid(v, cand_t)        the first returned atomic, if any (issued for the EX_U edges before any result is waited for)
//   tok2_t mid2(v, cand, tok)    the second one (the dedup mark of a successful relaxation), likewise
//   void   post(u, v, cand, tok, tok2)  the queue appends (wave-convergent: every lane that entered the step calls it EX_U times)
#ifndef CGA_EX_U
#define CGA_EX_U 4
#endif
constexpr int EX_U = CGA_EX_U;
template <typename Keep, typename F>
__device__ __forceinline__ void expand_frontier_mlp(int32_t const* q, int64_t n, int32_t const* offsets, int32_t const* indices, int32_t* bigq,
                                                    counters_t* cnt, Keep keep, F& f, int32_t big_deg = BIG_DEG, int32_t const* row_end = nullptr)
{
  __shared__ uint32_t s_scan[TV_WAVES][64];
  __shared__ eoff_t s_beg[TV_WAVES][64];
  __shared__ int32_t s_u[TV_WAVES][64];
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t const gwave  = (int64_t)blockIdx.x * TV_WAVES + wave;
  int64_t const nwaves = (int64_t)gridDim.x * TV_WAVES;
  unsigned long long inspected = 0;
  for (int64_t base = gwave * 64; base < n; base += nwaves * 64) {
    int64_t i = base + lane;
    int32_t u = -1, deg = 0;
    eoff_t beg = 0;
    if (i < n) {
      u = q ? q[i] : (int32_t)i;
      if (keep(u)) { beg = eoff(offsets, u); deg = (int32_t)((row_end ? eoff(row_end, u) : eoff(offsets, u + 1)) - beg); } else { u = -1; }
    }
    bool big = deg >= big_deg;
    if (big) {
      uint32_t nseg = ((uint32_t)deg + BIG_SEG - 1) / BIG_SEG;
      uint32_t at   = atomicAdd(&cnt->n_big, nseg);
      for (uint32_t sgm = 0; sgm < nseg; ++sgm) { bigq[2 * (at + sgm)] = u; bigq[2 * (at + sgm) + 1] = (int32_t)sgm; }
    }
    // whole-wave rows
    uint64_t mid = __ballot(deg >= 64 && !big);
    while (mid) {
      int src     = __ffsll((unsigned long long)mid) - 1;
      mid &= mid - 1;
      int32_t uu = __shfl(u, src), d = __shfl(deg, src);
      eoff_t const b = (eoff_t)__shfl((int)beg, src);
      for (int32_t p0 = lane; p0 < d; p0 += 64 * EX_U) {
        int32_t v[EX_U];
        typename F::cand_t c[EX_U];
        typename F::tok_t tk[EX_U];
        typename F::tok2_t tk2[EX_U];
#pragma unroll
        for (int k = 0; k < EX_U; ++k) { int32_t const p = p0 + 64 * k; v[k] = p < d ? indices[b + (eoff_t)p] : -1; }
#pragma unroll
        for (int k = 0; k < EX_U; ++k) { int32_t const p = p0 + 64 * k; c[k] = f.pre(uu, v[k], b + (eoff_t)(p < d ? p : 0)); }
#pragma unroll
        for (int k = 0; k < EX_U; ++k) tk[k] = f.mid(v[k], c[k]);
#pragma unroll
        for (int k = 0; k < EX_U; ++k) tk2[k] = f.mid2(v[k], c[k], tk[k]);
#pragma unroll
        for (int k = 0; k < EX_U; ++k) f.post(uu, v[k], c[k], tk[k], tk2[k]);
      }
      inspected += (lane == 0) ? (unsigned long long)d : 0ull;
    }
    // flattened small rows
    uint32_t sd = (deg < 64) ? (uint32_t)deg : 0u, total;
    uint32_t ex = wave_excl_scan(sd, lane, &total);
    s_scan[wave][lane] = ex;
    s_beg[wave][lane]  = beg;
    s_u[wave][lane]    = u;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t t0 = lane; t0 < total; t0 += 64 * EX_U) {
      int32_t v[EX_U], uu[EX_U];
      eoff_t pp[EX_U];
      typename F::cand_t c[EX_U];
      typename F::tok_t tk[EX_U];
      typename F::tok2_t tk2[EX_U];
#pragma unroll
      for (int k = 0; k < EX_U; ++k) {
        uint32_t const t = t0 + 64u * (uint32_t)k;
        v[k] = -1; uu[k] = 0; pp[k] = 0;  // (loadable stand-ins for "no edge": vertex 0, edge position 0 -- this loop runs only when edges exist)
        if (t < total) {
          int lo = 0, hi = 63;
#pragma unroll
          for (int st = 0; st < 6; ++st) {
            int m = (lo + hi + 1) >> 1;
            if (s_scan[wave][m] <= t) lo = m; else hi = m - 1;
          }
          pp[k] = s_beg[wave][lo] + (t - s_scan[wave][lo]);
          uu[k] = s_u[wave][lo];
          v[k]  = indices[pp[k]];
        }
      }
#pragma unroll
      for (int k = 0; k < EX_U; ++k) c[k] = f.pre(uu[k], v[k], pp[k]);
#pragma unroll
      for (int k = 0; k < EX_U; ++k) tk[k] = f.mid(v[k], c[k]);
#pragma unroll
      for (int k = 0; k < EX_U; ++k) tk2[k] = f.mid2(v[k], c[k], tk[k]);
#pragma unroll
      for (int k = 0; k < EX_U; ++k) f.post(uu[k], v[k], c[k], tk[k], tk2[k]);
    }
    __builtin_amdgcn_wave_barrier();
    inspected += (lane == 0) ? (unsigned long long)total : 0ull;
  }
  if (lane == 0 && inspected) atomicAdd(&cnt_replica(cnt)->edges, inspected);
}

template <typename F>
__device__ __forceinline__ void expand_big_mlp(int32_t const* bigq, int32_t const* offsets, int32_t const* indices, counters_t* cnt, F& f,
                                               int32_t const* row_end = nullptr)
{
  uint32_t const nseg = cnt->n_big;
  unsigned long long inspected = 0;
  for (uint32_t k = blockIdx.x; k < nseg; k += gridDim.x) {
    int32_t const u = bigq[2 * k], sgm = bigq[2 * k + 1];
    eoff_t const row_b = eoff(offsets, u), row_e = row_end ? eoff(row_end, u) : eoff(offsets, u + 1);
    eoff_t const b = row_b + (eoff_t)sgm * (eoff_t)BIG_SEG;
    eoff_t const len = min(row_e - b, (eoff_t)BIG_SEG);
    for (eoff_t p0 = threadIdx.x; p0 < len; p0 += (eoff_t)blockDim.x * EX_U) {
      int32_t v[EX_U];
      typename F::cand_t c[EX_U];
      typename F::tok_t tk[EX_U];
      typename F::tok2_t tk2[EX_U];
#pragma unroll
      for (int i = 0; i < EX_U; ++i) { eoff_t const p = p0 + (eoff_t)i * blockDim.x; v[i] = p < len ? indices[b + p] : -1; }
#pragma unroll
      for (int i = 0; i < EX_U; ++i) { eoff_t const p = p0 + (eoff_t)i * blockDim.x; c[i] = f.pre(u, v[i], b + (p < len ? p : 0)); }
#pragma unroll
      for (int i = 0; i < EX_U; ++i) tk[i] = f.mid(v[i], c[i]);
#pragma unroll
      for (int i = 0; i < EX_U; ++i) tk2[i] = f.mid2(v[i], c[i], tk[i]);
#pragma unroll
      for (int i = 0; i < EX_U; ++i) f.post(u, v[i], c[i], tk[i], tk2[i]);
    }
    if (threadIdx.x == 0) inspected += (unsigned long long)len;
  }
  if (threadIdx.x == 0 && inspected) atomicAdd(&cnt_replica(cnt)->edges, inspected);
}

inline size_t big_queue_entries(int64_t ne) { return (size_t)(2 * (ne / BIG_DEG_NARROW + ne / BIG_SEG + 64)); }
// largest edge count the traversal kernels address (32-bit unsigned positions, 2048 words of tail padding)
constexpr int64_t kMaxTraversalEdges = ((int64_t)1 << 32) - 4096;

// narrow frontier (fewer vertices than the chip has wavefront slots): defer every row a wavefront would otherwise walk alone
inline int32_t big_deg_for(handle_t const& h, int64_t n) { return n < (int64_t)h.num_cus * 64 ? BIG_DEG_NARROW : BIG_DEG; }

inline int expand_grid(handle_t const& h, int64_t n)
{
  int64_t waves = (n + 63) / 64;
  int64_t g     = (waves + TV_WAVES - 1) / TV_WAVES;
  int64_t cap   = (int64_t)h.num_cus * 8;
  return (int)std::max<int64_t>(1, std::min(g, cap));
}

}  // namespace
}  // namespace cga
