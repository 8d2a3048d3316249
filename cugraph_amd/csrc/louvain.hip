// cugraph_louvain.
//
// Replaces:
//   cugraph_louvain, cugraph_hierarchical_clustering_result_*   cpp/src/c_api/louvain.cpp:24-135, hierarchical_clustering_result.cpp
//   cugraph::louvain / detail::louvain                          cpp/src/community/louvain_impl.cuh:40-287
//   compute_modularity, update_clustering_by_delta_modularity,  cpp/src/community/detail/common_methods.cuh:52-479
//   compute_cluster_keys_and_values, graph_contraction            (over per_v_transform_reduce_dst_key_aggregated_outgoing_e, 1129 LoC
//                                                                  on cuco hash maps, and coarsen_graph_impl.cuh)
//   flatten_dendrogram                                           cpp/src/community/flatten_dendrogram.hpp:21-51
//
// The reference's algorithm with rng_state = nullopt is deterministic: synchronous local moving (every vertex picks the
// neighbouring cluster with the largest modularity gain, ties to the smaller cluster id, and moves only "up" or only "down"
// in alternate sweeps), a modularity test per sweep, contraction per level.  Here the key aggregation is a stable radix sort
// of the level's edges by (source, cluster of destination) followed by segment sums -- no hash maps -- and the sums do not depend
// on any execution order (below), so the result is bit-reproducible and equal to the oracle's (oracle/oracle.py: louvain,
// oracle.c: orc_louvain).  The gains are fp64, expressions in the reference's operation order, contraction off.
// Cluster labels of a contracted level = rank of the old label among the labels in use (the reference's labels are whatever
// its coarsen_graph renumbering assigns; its two C-API goldens come out identically, labels included).
// Scale (round 2): the per-vertex search for the best move is FLAT over the sorted edges (k_segment_sums / k_segment_best below):
// no loop over a vertex's row anywhere, so hub rows of 10^5 edges cost what their edges cost.  Segment sums, cluster weights and
// coarse edge weights are accumulated as 64-bit FIXED POINT with integer atomics (scale 2^s, s chosen from the total edge weight
// so nothing can overflow): integer addition is associative, so the result does not depend on the order the atomics land in,
// and for weights that are multiples of 2^-s -- integers, and every fp32 weight of moderate range -- it is exact, i.e. equal to
// the sequential fp64 sum the oracle forms.  The modularity reductions run over fixed 64 Ki-element chunks + one fixed-order fold.
// One radix sort of all edges per sweep except the first of a level, whose keys are the stored edge order (re-sorting only the
// rows whose neighbours moved is the next step).
#pragma clang fp contract(off)
#include "common.hpp"
#include "comm.hpp"
#include "mg_graph.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "cugraph_c/community_algorithms.h"

namespace cga {

struct clustering_result_t {  // c_api/hierarchical_clustering_result.hpp
  double modularity{0};
  device_array_t* vertices{nullptr};
  device_array_t* clusters{nullptr};
  ~clustering_result_t() { delete vertices; delete clusters; }
};

namespace {

int bits_of_u(uint64_t max_value)
{
  int b = 0;
  while (b < 64 && (max_value >> b) != 0) ++b;
  return b < 1 ? 1 : b;
}

#define LV_LOOP(i, n)                                                        \
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride_ = (int64_t)gridDim.x * blockDim.x; i < (n); i += stride_)

__global__ void k_expand_src(int32_t const* offsets, int64_t nv, int32_t* src)
{  // one wavefront per row (long rows are striped across the lanes)
  int64_t const wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int const lane = threadIdx.x & 63;
  for (int64_t v = wave; v < nv; v += nwaves) {
    uint32_t const b = (uint32_t)offsets[v], e = (uint32_t)offsets[v + 1];  // (positions: unsigned words)
    for (int64_t p = (int64_t)b + lane; p < (int64_t)e; p += 64) src[p] = (int32_t)v;
  }
}
template <typename WT>
__global__ void k_to_double(WT const* w, int64_t n, double* out) { LV_LOOP(i, n) out[i] = w ? (double)w[i] : 1.0; }

// keys (hi[i] or map_hi[hi[i]]) << shift | (lo[i] or map_lo[lo[i]]), payload i  (shift = bits of the low part: the significant
// bits are contiguous and one radix sort over 2 * shift bits orders the pairs)
__global__ void k_pair_keys(int32_t const* hi, int32_t const* map_hi, int32_t const* lo, int32_t const* map_lo, int64_t n, int shift, uint64_t* keys,
                            uint32_t* vals)
{
  LV_LOOP(i, n)
  {
    uint32_t const a = (uint32_t)(map_hi ? map_hi[hi[i]] : hi[i]), b = (uint32_t)(map_lo ? map_lo[lo[i]] : lo[i]);
    keys[i] = ((uint64_t)a << shift) | b;
    vals[i] = (uint32_t)i;
  }
}
__global__ void k_heads(uint64_t const* keys, int64_t n, uint32_t* head) { LV_LOOP(i, n) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u; }

// inclusive segmented scan over the 64 lanes (Hillis-Steele); `head` marks the first element of a segment, `open` tells on
// return whether the lane's segment started before lane 0 (no head at or below the lane)
template <typename T>
__device__ __forceinline__ T seg_scan64(T val, bool head, int lane, bool& open)
{
  unsigned f = head ? 1u : 0u;
  for (int o = 1; o < 64; o <<= 1) {
    T const t         = __shfl_up(val, o);
    unsigned const tf = __shfl_up(f, o);
    if (lane >= o && !f) { val += t; f |= tf; }
  }
  open = f == 0;
  return val;
}
// vertex weights k[v] = sum of the vertex's edge weights, in fixed point (kfix, for the cluster-weight atomics) and as double:
// flat over the edges (grouped by source), a wavefront reduces its 64 entries by source and adds one value per (wavefront, source)
__global__ void k_vertex_weights(int32_t const* src, double const* w, int64_t ne, double scale, unsigned long long* kfix)
{
  int const lane       = threadIdx.x & 63;
  int64_t const wave   = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t const nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i0 = wave * 64; i0 < ne; i0 += nwaves * 64) {
    int64_t const i  = i0 + lane;
    bool const valid = i < ne;
    int32_t const v     = valid ? src[i] : -1;
    int32_t const vprev = __shfl_up(v, 1), vnext = __shfl_down(v, 1);
    long long const wf  = valid ? __double2ll_rn(w[i] * scale) : 0;
    bool open;
    long long const sum = seg_scan64(wf, lane == 0 || v != vprev, lane, open);
    if (valid && (lane == 63 || v != vnext)) atomicAdd(&kfix[v], (unsigned long long)sum);
  }
}

// The best move of every vertex, flat over the edges sorted by (vertex, cluster of destination, stored order) -- no per-row loop,
// so a hub row costs what its edges cost.  detail::key_aggregated_edge_op_t + reduce_op_t (common_methods.cuh:70-125) after the
// old_cluster_sum / cluster_subtract pass (:335-362):
//   k_segment_sums   a wavefront takes 64 consecutive entries; a SEGMENT is a run of equal keys (vertex, cluster).  Segmented scan
//                    of the fixed-point weights; every piece of a segment that ends in the wavefront is added to segfix[position
//                    of the segment's first entry] (a plain store when the whole segment lies in the wavefront; the first entry
//                    of a segment that started in an earlier wavefront is found by a galloping search in the sorted keys).  The
//                    same pieces feed selffix[v] (weight into the vertex's own cluster) and subfix[v] (self-loops).
//   k_segment_best<0> at every segment head: the modularity gain of moving the vertex to that cluster; maximum per vertex
//                    (positive doubles order like their bit patterns: integer atomicMax, reduced per wavefront first)
//   k_segment_best<1> the smallest cluster id among the segments that attain the maximum (atomicMin)
struct lv_flat_args {
  uint64_t const* keys; uint32_t const* perm; int32_t const* dst; double const* w; int32_t const* c;
  double const* k; double const* a; double m, resolution, scale, inv_scale; int64_t ne; int vb;
  unsigned long long* segfix;     // [ne], indexed by the position of a segment's first entry; zero on entry
  unsigned long long* selffix;    // [nv] zero on entry
  unsigned long long* subfix;     // [nv] zero on entry
  unsigned long long* best_bits;  // [nv] zero on entry
  int32_t* best_c;                // [nv] 0x7f7f7f7f on entry
};
__device__ __forceinline__ double lv_delta(double new_sum, double old_sum, double a_new, double a_old, double kk, double m, double resolution)
{
  return 2.0 * (((new_sum - old_sum) / m) - resolution * (a_new * kk - a_old * kk + kk * kk) / (m * m));
}
// first position q <= hi with keys[q] == key, given keys[hi] == key (keys ascending)
__device__ __forceinline__ int64_t lv_first_equal(uint64_t const* keys, uint64_t key, int64_t hi)
{
  int64_t lo = hi, step = 64;
  while (lo - step >= 0 && keys[lo - step] == key) { lo -= step; step <<= 1; }
  int64_t below = lo - step >= 0 ? lo - step : -1;  // keys[below] != key (or below == -1); keys[lo] == key
  while (lo - below > 1) {
    int64_t const mid = below + (lo - below) / 2;
    if (keys[mid] == key) lo = mid; else below = mid;
  }
  return lo;
}
__global__ void k_segment_sums(lv_flat_args A)
{
  int const lane       = threadIdx.x & 63;
  int64_t const wave   = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t const nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  uint64_t const cmask = (1ull << A.vb) - 1ull;
  for (int64_t i0 = wave * 64; i0 < A.ne; i0 += nwaves * 64) {
    int64_t const p  = i0 + lane;
    bool const valid = p < A.ne;
    uint64_t key = ~0ull, prev = ~1ull, next = ~2ull;  // real keys have at most 62 significant bits
    if (valid) {
      key = A.keys[p];
      if (p > 0) prev = A.keys[p - 1];
      if (p + 1 < A.ne) next = A.keys[p + 1];
    }
    bool const head = key != prev;  // (invalid lanes are heads of their own)
    bool const end  = valid && key != next;
    int32_t const v  = (int32_t)(key >> A.vb);
    int32_t const cl = (int32_t)(key & cmask);
    long long wf = 0;
    if (valid) {
      uint32_t const ep = A.perm[p];
      wf = __double2ll_rn(A.w[ep] * A.scale);
      if (A.dst[ep] == v) atomicAdd(&A.subfix[v], (unsigned long long)wf);  // self-loop
    }
    bool open;
    long long const sum = seg_scan64(wf, head || lane == 0, lane, open);
    int hl = head ? lane : -1;  // lane of the piece's first entry when the segment starts in this wavefront
    for (int o = 1; o < 64; o <<= 1) {
      int const t = __shfl_up(hl, o);
      if (lane >= o && t > hl) hl = t;
    }
    if (valid && (end || lane == 63)) {  // a piece of a segment ends here
      int64_t const hp = hl >= 0 ? i0 + hl : lv_first_equal(A.keys, key, i0 - 1);
      if (hl >= 0 && end) A.segfix[hp] = (unsigned long long)sum;  // the whole segment lies in this wavefront
      else atomicAdd(&A.segfix[hp], (unsigned long long)sum);
      if (cl == A.c[v]) atomicAdd(&A.selffix[v], (unsigned long long)sum);
    }
  }
}
template <int PHASE>
__global__ void k_segment_best(lv_flat_args A)
{
  int const lane       = threadIdx.x & 63;
  int64_t const wave   = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t const nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  uint64_t const cmask = (1ull << A.vb) - 1ull;
  for (int64_t i0 = wave * 64; i0 < A.ne; i0 += nwaves * 64) {
    int64_t const p  = i0 + lane;
    bool const valid = p < A.ne;
    uint64_t key = ~0ull, prev = ~1ull;
    if (valid) {
      key = A.keys[p];
      if (p > 0) prev = A.keys[p - 1];
    }
    bool const head  = valid && key != prev;
    int32_t const v  = valid ? (int32_t)(key >> A.vb) : -1;
    int32_t const cl = (int32_t)(key & cmask);
    unsigned long long bits = 0;
    if (head) {
      int32_t const cv     = A.c[v];
      double const s       = (double)(long long)A.segfix[p] * A.inv_scale;
      double const sub     = (double)(long long)A.subfix[v] * A.inv_scale;
      double const old_sum = (double)(long long)(A.selffix[v] - A.subfix[v]) * A.inv_scale;
      double const new_sum = cl == cv ? s - sub : s;
      double const delta   = lv_delta(new_sum, old_sum, A.a[cl], A.a[cv], A.k[v], A.m, A.resolution);
      if (delta > 0.0) bits = (unsigned long long)__double_as_longlong(delta);
    }
    if (PHASE == 0) {
      // maximum over the lanes of one vertex (consecutive lanes), then one atomic per (wavefront, vertex)
      int32_t const vprev = __shfl_up(v, 1), vnext = __shfl_down(v, 1);
      bool vhead = lane == 0 || v != vprev;
      unsigned long long mx = bits;
      unsigned f = vhead ? 1u : 0u;
      for (int o = 1; o < 64; o <<= 1) {
        unsigned long long const t = __shfl_up(mx, o);
        unsigned const tf          = __shfl_up(f, o);
        if (lane >= o && !f) { mx = t > mx ? t : mx; f |= tf; }
      }
      bool const vend = lane == 63 || v != vnext;
      if (valid && vend && mx) atomicMax(&A.best_bits[v], mx);
    } else {
      if (bits && bits == A.best_bits[v]) atomicMin(&A.best_c[v], cl);
    }
  }
}
// ---------------------------------------------------------------------------------------------------------------------------
// Round 3: the best move of every vertex WITHOUT sorting, for the rows of at most LVH_B edges.
// A workgroup takes the rows whose first edge lies in one window of LVH_B consecutive edge positions (fewer than 2 * LVH_B
// edges: every row but the last ends inside the window, the last has at most LVH_B edges; a longer row -- a hub, which goes
// through the sorted path below -- can only be the LAST row that starts in a window, so the edges of a chunk are contiguous).
// The weight from a vertex into every neighbouring cluster is accumulated in an LDS hash table keyed by (row slot, cluster) with
// 64-bit fixed-point integer atomics -- the same sums as the sorted path, bit for bit, whatever order the edges arrive in --,
// (round 4 measured the gains formed per occupied slot -- per distinct pair -- instead of per edge: 0.0757 against 0.0745 s at RMAT-22, no gain;)
// then every edge looks its (row, cluster) sum up, forms the modularity gain with the reference's expression
// (common_methods.cuh:70-125) and the row's maximum / smallest cluster among the maxima are reduced in LDS.  One pass over
// (destination, weight, cluster of destination) per edge instead of key construction + 6 radix passes + three segment passes.
// The row slot of a vertex is the position of its first edge relative to the window (unique per non-empty row, < LVH_B).
#ifndef LVH_THREADS_N
#define LVH_THREADS_N 1024  // 256 -> 1024 threads per workgroup (two workgroups of 56 KB LDS per CU either way): 0.0844 -> 0.078 s at RMAT-22 (round 4)
#endif
constexpr int LVH_B = 512, LVH_CAP = 2 * LVH_B, LVH_SLOTS = 2048, LVH_THREADS = LVH_THREADS_N;
constexpr unsigned long long LVH_EMPTY = ~0ull;
struct lv_hash_args {
  int32_t const* src; int32_t const* dst; double const* w; double const* k;
  int4 const* ca;                 // [nv] per vertex u: (c[u], -, a[c[u]] as two words) -- ONE 16-byte gather per edge instead of c[u] and then a[c[u]] (k_lv_pack_ca, once per sweep)
  double m, resolution, scale, inv_scale; int64_t ne;
  unsigned long long* best_bits;  // [nv]
  int32_t* best_c;                // [nv]
  unsigned long long* ipart;      // [chunks] weight of the chunk's rows' edges that stay inside their cluster (fixed point): the modularity's first term.  One word per
                                  // workgroup, summed afterwards (round 6: was one atomic per workgroup on ONE address -- 127 K of them per sweep at RMAT-22)
  // what depends on the level's graph only (round 5: computed once per level by k_lv_chunk_prep instead of by every sweep's workgroups --
  // a chain of four dependent loads on ONE thread before a workgroup could start, and src -> off per edge)
  long long const* range;         // [chunks][2] edge range [p0, p1) of every chunk (p1 <= p0: nothing)
  uint16_t const* rs;             // [ne] row slot of every edge = position of its row's first edge inside that row's window (off[src] mod LVH_B); bit 15: the edge is a self-loop
  int64_t chunk0;                 // first chunk of this launch (a launch holds fewer than 2^32 threads: levels of more than 2^31 edges take several)
};
__device__ __forceinline__ uint32_t lvh_slot(uint32_t rs, uint32_t cl) { return ((rs * 0x9E3779B1u) ^ (cl * 0x85EBCA6Bu) ^ (cl >> 15)) & (LVH_SLOTS - 1); }
static_assert((LVH_B & (LVH_B - 1)) == 0 && LVH_B <= 0x8000, "the row slot is off[v] mod LVH_B and shares 16 bits with the self-loop flag");
static_assert(LVH_CAP % LVH_THREADS == 0, "a chunk's edges live in registers between the passes: LVH_CAP / LVH_THREADS per thread");
// once per level: the edge range of every chunk (the rule of round 3, see above) and the row slot of every edge
__global__ void k_lv_chunk_prep(int32_t const* src, int32_t const* dst, uint32_t const* off, int64_t ne, long long* range, uint16_t* rs)
{
  int64_t const t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  int64_t const n_chunks = (ne + LVH_B - 1) / LVH_B;
  for (int64_t b = t; b < n_chunks; b += stride) {
    int64_t const e_lo = b * (int64_t)LVH_B, e_hi = e_lo + LVH_B < ne ? e_lo + LVH_B : ne;
    int64_t p0 = e_lo, p1 = e_lo;
    if (e_lo > 0 && src[e_lo - 1] == src[e_lo]) p0 = off[src[e_lo] + 1];  // the window opens inside a row of an earlier chunk (or a hub)
    if (p0 < e_hi) {
      int32_t const vl = src[e_hi - 1];  // the last row that starts in the window
      int64_t const bb = off[vl], ee = off[vl + 1];
      p1 = ee - bb > LVH_B ? bb : ee;    // a hub can only be the last row of a window
    }
    range[2 * b]     = p0;
    range[2 * b + 1] = p1 > p0 ? p1 : p0;
  }
  for (int64_t e = t; e < ne; e += stride) rs[e] = (uint16_t)(((uint32_t)off[src[e]] & (uint32_t)(LVH_B - 1)) | (src[e] == dst[e] ? 0x8000u : 0u));
}
// once per sweep: the destination's cluster and that cluster's weight side by side (nv gathers of a[] instead of one per edge)
__global__ void k_lv_pack_ca(int32_t const* c, double const* a, int64_t nv, int4* ca)
{
  LV_LOOP(v, nv)
  {
    int32_t const cl = c[v];
    long long const ab = __double_as_longlong(a[cl]);
    ca[v] = make_int4(cl, 0, (int32_t)(uint32_t)(unsigned long long)ab, (int32_t)(uint32_t)((unsigned long long)ab >> 32));
  }
}
__device__ __forceinline__ double lv_ca_weight(int4 const& q) { return __longlong_as_double((long long)(((unsigned long long)(uint32_t)q.w << 32) | (unsigned long long)(uint32_t)q.z)); }
// LVH_E = LVH_CAP / LVH_THREADS edges per thread (a chunk has fewer than 2 * LVH_B edges): an edge's cluster, table slot, cluster weight and gain
// stay in registers between the passes; global memory is touched once per edge -- (row slot, destination, weight) streamed, ONE 16-byte gather --
// and once per row (source, the row's own pair, its vertex weight).  Round 6: was two dependent gathers per edge (c[dst], then a[c[dst]] in
// the second pass) and a re-probe of the table per edge.
constexpr int LVH_E = LVH_CAP / LVH_THREADS;
__global__ void __launch_bounds__(LVH_THREADS) k_lv_hash_chunks(lv_hash_args A)
{
  __shared__ unsigned long long s_key[LVH_SLOTS], s_sum[LVH_SLOTS];
  __shared__ unsigned long long s_sub[LVH_B], s_best[LVH_B];
  __shared__ double s_k[LVH_B], s_acv[LVH_B];  // per row: the vertex's weight, the weight of its cluster (gathered once per row, by the row's first edge)
  __shared__ int32_t s_bestc[LVH_B], s_cv[LVH_B];
  __shared__ unsigned long long s_int;
  int const tid = threadIdx.x;
  int64_t const chunk = A.chunk0 + blockIdx.x;
  int64_t const e_lo = chunk * (int64_t)LVH_B;
  int64_t const p0   = A.range[2 * chunk];
  int const n        = (int)(A.range[2 * chunk + 1] - p0);  // < 2 * LVH_B
  if (n <= 0) { if (tid == 0) A.ipart[chunk] = 0; return; }
  uint32_t rsf[LVH_E];
  int32_t u[LVH_E];
  double w[LVH_E];
#pragma unroll
  for (int j = 0; j < LVH_E; ++j) {  // (in flight while the tables are cleared)
    int const i = tid + j * LVH_THREADS;
    rsf[j] = 0; u[j] = 0; w[j] = 0.0;
    // (streamed once per sweep: non-temporal, the lines the gathers below want stay in L2 -- 1 % of the call, profiles/r6x2)
    if (i < n) { rsf[j] = __builtin_nontemporal_load(&A.rs[p0 + i]); u[j] = __builtin_nontemporal_load(&A.dst[p0 + i]); w[j] = __builtin_nontemporal_load(&A.w[p0 + i]); }
  }
  if (tid == 0) s_int = 0;
  for (int i = tid; i < LVH_SLOTS; i += LVH_THREADS) { s_key[i] = LVH_EMPTY; s_sum[i] = 0; }
  for (int i = tid; i < LVH_B; i += LVH_THREADS) { s_sub[i] = 0; s_best[i] = 0; s_bestc[i] = 0x7f7f7f7f; }
  int4 cau[LVH_E], cav[LVH_E];
  int32_t v[LVH_E];
  double kv[LVH_E];
  bool first[LVH_E];
#pragma unroll
  for (int j = 0; j < LVH_E; ++j) {
    int const i = tid + j * LVH_THREADS;
    cau[j] = make_int4(0, 0, 0, 0); cav[j] = cau[j]; v[j] = -1; kv[j] = 0.0;
    first[j] = i < n && (uint32_t)(p0 + i - e_lo) == (rsf[j] & 0x7fffu);  // the row's first edge (rows of a chunk start inside its window)
    if (i < n) cau[j] = A.ca[u[j]];
    if (first[j]) v[j] = A.src[p0 + i];
  }
#pragma unroll
  for (int j = 0; j < LVH_E; ++j)
    if (first[j]) { cav[j] = A.ca[v[j]]; kv[j] = A.k[v[j]]; }
  __syncthreads();
  // pass 1: (row slot, cluster of destination) -> sum of weights; self-loops per row; the row's own operands of the gain
  uint32_t slot[LVH_E];
#pragma unroll
  for (int j = 0; j < LVH_E; ++j) {
    uint32_t const rs = rsf[j] & 0x7fffu, cl = (uint32_t)cau[j].x;
    slot[j] = lvh_slot(rs, cl);
    if (tid + j * LVH_THREADS < n) {
      unsigned long long const wf  = (unsigned long long)__double2ll_rn(w[j] * A.scale);
      unsigned long long const key = ((unsigned long long)rs << 32) | cl;
      for (;;) {
        unsigned long long const old = atomicCAS(&s_key[slot[j]], LVH_EMPTY, key);
        if (old == LVH_EMPTY || old == key) break;
        slot[j] = (slot[j] + 1) & (LVH_SLOTS - 1);
      }
      atomicAdd(&s_sum[slot[j]], wf);
      if (rsf[j] & 0x8000u) atomicAdd(&s_sub[rs], wf);
    }
    if (first[j]) { s_cv[rs] = cav[j].x; s_k[rs] = kv[j]; s_acv[rs] = lv_ca_weight(cav[j]); }
  }
  __syncthreads();
  auto lookup = [&](uint32_t rs2, uint32_t cl2) -> unsigned long long {  // 0 when the row has no edge into the cluster
    unsigned long long const key = ((unsigned long long)rs2 << 32) | cl2;
    uint32_t sl = lvh_slot(rs2, cl2);
    for (;;) {
      unsigned long long const k2 = s_key[sl];
      if (k2 == key) return s_sum[sl];
      if (k2 == LVH_EMPTY) return 0ull;
      sl = (sl + 1) & (LVH_SLOTS - 1);
    }
  };
  // pass 2: the gain of every edge's (row, cluster) pair (pairs that occur on several edges are evaluated as often: same value)
  unsigned long long bits[LVH_E];
#pragma unroll
  for (int j = 0; j < LVH_E; ++j) {
    bits[j] = 0;
    if (tid + j * LVH_THREADS < n) {
      uint32_t const rs = rsf[j] & 0x7fffu, cl = (uint32_t)cau[j].x;
      int32_t const cv = s_cv[rs];
      unsigned long long const sfix = s_sum[slot[j]];
      unsigned long long const self = (int32_t)cl == cv ? sfix : lookup(rs, (uint32_t)cv);
      unsigned long long const subf = s_sub[rs];
      double const s       = (double)(long long)sfix * A.inv_scale;
      double const sub     = (double)(long long)subf * A.inv_scale;
      double const old_sum = (double)(long long)(self - subf) * A.inv_scale;
      double const new_sum = (int32_t)cl == cv ? s - sub : s;
      double const delta   = lv_delta(new_sum, old_sum, lv_ca_weight(cau[j]), s_acv[rs], s_k[rs], A.m, A.resolution);
      bits[j] = delta > 0.0 ? (unsigned long long)__double_as_longlong(delta) : 0ull;
      if (bits[j]) atomicMax(&s_best[rs], bits[j]);  // positive doubles order like their bit patterns
    }
  }
  __syncthreads();
  // pass 3: the smallest cluster id among the pairs that attain the row's maximum (the reference's tie rule)
#pragma unroll
  for (int j = 0; j < LVH_E; ++j) {
    uint32_t const rs = rsf[j] & 0x7fffu;
    if (bits[j] && bits[j] == s_best[rs]) atomicMin(&s_bestc[rs], cau[j].x);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < LVH_E; ++j)
    if (first[j]) {
      uint32_t const rs = rsf[j] & 0x7fffu;
      unsigned long long const best = s_best[rs];
      A.best_bits[v[j]] = best;
      A.best_c[v[j]]    = best ? s_bestc[rs] : 0x7f7f7f7f;
      unsigned long long const self = lookup(rs, (uint32_t)cav[j].x);
      if (self) atomicAdd(&s_int, self);
    }
  __syncthreads();
  if (tid == 0) A.ipart[chunk] = s_int;
}
// Round 4: rows of LVH_B < degree <= LVM_MAX edges ("mid rows": a quarter of the edges of RMAT-22 that used to take the sorted path with
// the hubs): ONE workgroup per row, the row's (cluster -> weight) sums in an LDS open-addressing table keyed by the cluster alone (at most
// degree distinct keys in a table of at least twice as many slots), fixed-point integer atomics -- the same integers whatever the order --,
// then the table's slots are walked once: gain per occupied slot with the same expression and the same operands as the other two paths,
// maximum gain and smallest cluster among the maxima reduced over the workgroup.  One pass over (destination, weight, cluster) per edge.
constexpr int LVM_MAX = 4096, LVM_THREADS = 512;
struct lv_mid_args {
  int32_t const* rows; int32_t n_rows;  // vertices with LVH_B < degree <= max_deg of this launch, any order
  int32_t const* dst; uint32_t const* off; double const* w; int32_t const* c; double const* k; double const* a;
  double m, resolution, scale, inv_scale;
  unsigned long long* best_bits; int32_t* best_c;
  unsigned long long* ifix;  // += the rows' weight into their own clusters
};
template <int SLOTS>
__global__ void __launch_bounds__(LVM_THREADS) k_lv_hash_rows(lv_mid_args A)
{
  // Round 6: a row's loads are issued together -- all of a thread's (destination, weight) pairs, then all their clusters, then all the weights of the
  // clusters in its slots -- and the NEXT row's header (vertex, offsets, cluster) is fetched while this row is worked on: a row used to be a chain of
  // eight dependent global round trips with the whole workgroup waiting on each.
  constexpr int EPT = SLOTS / 2 / LVM_THREADS;  // a launch's rows have at most SLOTS / 2 edges
  constexpr int SPT = SLOTS / LVM_THREADS;
  extern __shared__ unsigned long long lvm_smem[];
  unsigned long long* const s_sum = lvm_smem;                                  // [SLOTS]
  uint32_t* const s_key           = reinterpret_cast<uint32_t*>(s_sum + SLOTS);  // [SLOTS]
  __shared__ unsigned long long s_sub, s_red_bits[LVM_THREADS / 64];
  __shared__ int32_t s_red_c[LVM_THREADS / 64];
  int const tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto slot_of = [](uint32_t cl) { return ((cl * 0x9E3779B1u) ^ (cl >> 15)) & (uint32_t)(SLOTS - 1); };
  unsigned long long internal = 0;  // (the same value in every thread)
  int32_t v_n = 0, d_n = 0, cv_n = 0;
  uint32_t b_n = 0;
  if ((int)blockIdx.x < A.n_rows) { v_n = A.rows[blockIdx.x]; b_n = A.off[v_n]; d_n = (int32_t)(A.off[v_n + 1] - b_n); cv_n = A.c[v_n]; }
  for (int r = blockIdx.x; r < A.n_rows; r += gridDim.x) {
    int32_t const v = v_n, d = d_n, cv = cv_n;
    uint32_t const b = b_n;
    int32_t u[EPT];
    double w[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      int const i = tid + j * LVM_THREADS;
      u[j] = 0; w[j] = 0.0;
      if (i < d) { u[j] = A.dst[b + i]; w[j] = A.w[b + i]; }
    }
    double const a_old = A.a[cv], kk = A.k[v];
    if (r + (int)gridDim.x < A.n_rows) { v_n = A.rows[r + gridDim.x]; b_n = A.off[v_n]; d_n = (int32_t)(A.off[v_n + 1] - b_n); cv_n = A.c[v_n]; }
    for (int i = tid; i < SLOTS; i += LVM_THREADS) { s_key[i] = 0xFFFFFFFFu; s_sum[i] = 0; }
    if (tid == 0) s_sub = 0;
    uint32_t cl[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) cl[j] = tid + j * LVM_THREADS < d ? (uint32_t)A.c[u[j]] : 0u;
    __syncthreads();
    unsigned long long sub = 0;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      if (tid + j * LVM_THREADS < d) {
        unsigned long long const wf = (unsigned long long)__double2ll_rn(w[j] * A.scale);
        uint32_t slot = slot_of(cl[j]);
        for (;;) {
          uint32_t const old = atomicCAS(&s_key[slot], 0xFFFFFFFFu, cl[j]);
          if (old == 0xFFFFFFFFu || old == cl[j]) break;
          slot = (slot + 1) & (uint32_t)(SLOTS - 1);
        }
        atomicAdd(&s_sum[slot], wf);
        if (u[j] == v) sub += wf;
      }
    }
    if (sub) atomicAdd(&s_sub, sub);
    __syncthreads();
    uint32_t kcl[SPT];
    double acl[SPT];
#pragma unroll
    for (int j = 0; j < SPT; ++j) kcl[j] = s_key[tid + j * LVM_THREADS];
#pragma unroll
    for (int j = 0; j < SPT; ++j) acl[j] = kcl[j] != 0xFFFFFFFFu ? A.a[kcl[j]] : 0.0;
    unsigned long long self = 0;
    {
      uint32_t slot = slot_of((uint32_t)cv);
      for (;;) {
        uint32_t const k2 = s_key[slot];
        if (k2 == (uint32_t)cv) { self = s_sum[slot]; break; }
        if (k2 == 0xFFFFFFFFu) break;
        slot = (slot + 1) & (uint32_t)(SLOTS - 1);
      }
    }
    internal += self;
    unsigned long long const subf = s_sub;
    double const sub_d = (double)(long long)subf * A.inv_scale, old_sum = (double)(long long)(self - subf) * A.inv_scale;
    unsigned long long best = 0;
    int32_t best_c = 0x7f7f7f7f;
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      uint32_t const c2 = kcl[j];
      if (c2 == 0xFFFFFFFFu) continue;
      double const sd      = (double)(long long)s_sum[tid + j * LVM_THREADS] * A.inv_scale;
      double const new_sum = (int32_t)c2 == cv ? sd - sub_d : sd;
      double const delta   = lv_delta(new_sum, old_sum, acl[j], a_old, kk, A.m, A.resolution);
      unsigned long long const bits = delta > 0.0 ? (unsigned long long)__double_as_longlong(delta) : 0ull;
      if (bits > best || (bits == best && bits && (int32_t)c2 < best_c)) { best = bits; best_c = (int32_t)c2; }
    }
    for (int o = 32; o; o >>= 1) {
      unsigned long long const ob = __shfl_xor(best, o);
      int32_t const oc            = __shfl_xor(best_c, o);
      if (ob > best || (ob == best && oc < best_c)) { best = ob; best_c = oc; }
    }
    if (lane == 0) { s_red_bits[wave] = best; s_red_c[wave] = best_c; }
    __syncthreads();
    if (tid == 0) {
      for (int k2 = 1; k2 < LVM_THREADS / 64; ++k2)
        if (s_red_bits[k2] > best || (s_red_bits[k2] == best && s_red_c[k2] < best_c)) { best = s_red_bits[k2]; best_c = s_red_c[k2]; }
      A.best_bits[v] = best;
      A.best_c[v]    = best ? best_c : 0x7f7f7f7f;
    }
    __syncthreads();
  }
  if (tid == 0 && internal) atomicAdd(A.ifix, internal);
}
// Rows of more than LVM_MAX edges ("big rows"): R = ceil(degree / LVB_SHARE) work items per row; item r scans the WHOLE row and keeps the
// clusters whose range hash is r (a 1 / R share of the distinct clusters: <= LVB_SHARE expected in a table of LVB_SLOTS), so the table of
// a row of any length lives in LDS, at the price of reading the row R times (the items of a row run next to each other: the re-reads are
// L2 hits; 2.3 x the big rows' edges at RMAT-22).  Every item also accumulates the row's self-loop weight and the weight into the row's
// own cluster while it scans (each item needs both for the gains).  An item reduces its slots to (best gain, smallest cluster among
// them), raises best_bits[v] with an atomic maximum and keeps its pair; k_lv_big_ties then lets the items whose gain IS the row's maximum
// lower best_c[v] -- the same maximum and the same tie rule as the sorted path's k_segment_best<0/1>.
// A table that fills up (keys that defeat the range hash) raises *overflow and the item gives up: the caller repeats the sweep's
// evaluation with the sorted path for these rows.
// Table size and workgroup shape (round 6, last session, with the partition of the big rows' pairs in place: an item's cost is then its latency chain, not the
// pairs it reads, and smaller tables mean more items in flight per CU): 4096 slots x 512 threads (48 KB: three workgroups per CU) RMAT-26 1.24-1.25 s, RMAT-22
// 0.0626; 8192 x 1024 (one per CU; the default while every item scanned its whole row, where it won 1.33 against 1.36 s) 1.28 / 0.0625; 4096 x 1024 1.31-1.35 /
// 0.066-0.068; 4096 x 256 1.25 / 0.0647; 2048 x 256 1.26 / 0.065; 2048 x 512 1.26-1.28 / 0.0644 (profiles/r6bn, r6bo)
#ifndef LVB_SLOTS_N
#define LVB_SLOTS_N 4096
#endif
#ifndef LVB_THREADS_N
#define LVB_THREADS_N 512
#endif
constexpr int LVB_SLOTS = LVB_SLOTS_N, LVB_SHARE = LVB_SLOTS * 3 / 8, LVB_THREADS = LVB_THREADS_N;
struct lv_big_args {
  int4 const* items; int32_t n_items;  // (row, r, R, position of the row's first edge in the big rows' edge list)
  uint32_t const* ecl; unsigned long long const* ewf;  // per edge of that list: cluster of the destination (this sweep), fixed-point weight (this level)
  unsigned long long const* rowsub;                    // [nv] self-loop weight of a row (this level)
  uint32_t const* off; int32_t const* c; double const* k; double const* a;
  double m, resolution, scale, inv_scale;
  unsigned long long* best_bits; int32_t* best_c;
  unsigned long long* item_bits; int32_t* item_c;  // [n_items]
  uint32_t* overflow;
  unsigned long long* ifix;  // += the rows' weight into their own clusters (by a row's first item)
  uint32_t max_used;  // slots an item may occupy (an eighth stays free: probes stay short)
  uint32_t* cursor;   // next item (zero on entry): items are drawn dynamically, the longest rows' items first (k_lv_big_items is launched longest rows first)
  // this sweep's partition of every big row by item (k_lv_big_partition): item i owns entries [seg[i].x, seg[i].x + seg[i].y) of (pcl, pwf)
  uint2 const* seg; uint32_t const* pcl; unsigned long long const* pwf;
  unsigned long long const* rowself;  // [nv] weight of a big row into its own cluster (this sweep)
};
__device__ __forceinline__ uint32_t lvb_range(uint32_t cl, uint32_t R) { return (uint32_t)(((unsigned long long)(cl * 0x85EBCA6Bu + 0x27D4EB2Fu) * R) >> 32); }
template <bool PART>  // PART: the items read their own pairs from this sweep's partition (k_lv_big_partition); otherwise every item scans its whole row
__global__ void __launch_bounds__(LVB_THREADS) k_lv_hash_big(lv_big_args A)
{
  extern __shared__ unsigned long long lvm_smem[];
  unsigned long long* const s_sum = lvm_smem;                                      // [LVB_SLOTS]
  uint32_t* const s_key           = reinterpret_cast<uint32_t*>(s_sum + LVB_SLOTS);  // [LVB_SLOTS]
  __shared__ unsigned long long s_self, s_red_bits[LVB_THREADS / 64];
  __shared__ int32_t s_red_c[LVB_THREADS / 64];
  __shared__ uint32_t s_used, s_full, s_it;
  int const tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto slot_of = [](uint32_t cl) { return ((cl * 0x9E3779B1u) ^ (cl >> 15)) & (uint32_t)(LVB_SLOTS - 1); };
  // An item of a row of 855 K edges scans 280 times what an item of a row of 8 K edges scans (RMAT-26): items are drawn from a queue, longest rows first,
  // instead of being dealt out by index (round 6)
  for (;;) {
    if (tid == 0) s_it = atomicAdd(A.cursor, 1u);
    __syncthreads();
    int const it = (int)s_it;
    if (it >= A.n_items) break;  // (uniform)
    int4 const item = A.items[it];
    int32_t const v = item.x;
    uint32_t const r = (uint32_t)item.y, R = (uint32_t)item.z;
    for (int i = tid; i < LVB_SLOTS; i += LVB_THREADS) { s_key[i] = 0xFFFFFFFFu; s_sum[i] = 0; }
    if (tid == 0) { s_self = 0; s_used = 0; s_full = 0; }
    __syncthreads();
    int32_t const cv = A.c[v];
    // PART: k_lv_big_partition has grouped this sweep's (cluster, weight) pairs of every big row by item, and the item reads its own pairs only
    // (round 6, last session).  Otherwise every item scans the WHOLE row and keeps its range -- d R pairs read per row: 2.3 x the big rows' edges at RMAT-22,
    // where that is cheaper than partitioning them (0.0636 against 0.0668 s), 11 x at RMAT-26, where it is not (1.40 against 1.30 s); run_level decides by that factor
    constexpr bool part = PART;
    uint2 const sg  = part ? A.seg[it] : make_uint2((uint32_t)item.w, A.off[v + 1] - A.off[v]);
    int const d_it  = (int)sg.y;
    uint32_t const* const ecl           = (part ? A.pcl : A.ecl) + sg.x;
    unsigned long long const* const ewf = (part ? A.pwf : A.ewf) + sg.x;
    unsigned long long self = 0;
    constexpr int LVB_UNROLL = 4;  // (round 6: four (cluster, weight) pairs per thread in flight; the loop used to wait for each pair)
    for (int i0 = tid; i0 < d_it; i0 += LVB_THREADS * LVB_UNROLL) {
      uint32_t cls[LVB_UNROLL];
      unsigned long long wfs[LVB_UNROLL];
#pragma unroll
      for (int j = 0; j < LVB_UNROLL; ++j) {
        int const i = i0 + j * LVB_THREADS;
        cls[j] = 0; wfs[j] = 0;
        if (i < d_it) { cls[j] = ecl[i]; wfs[j] = ewf[i]; }
      }
#pragma unroll
      for (int j = 0; j < LVB_UNROLL; ++j) {
        if (i0 + j * LVB_THREADS >= d_it) continue;
        uint32_t const cl           = cls[j];
        unsigned long long const wf = wfs[j];
        if constexpr (!part) {
          if ((int32_t)cl == cv) self += wf;
          if (lvb_range(cl, R) != r) continue;
        }
        uint32_t slot = slot_of(cl);
        bool placed   = false;
        for (int probes = 0; probes < LVB_SLOTS; ++probes) {
          uint32_t const old = atomicCAS(&s_key[slot], 0xFFFFFFFFu, cl);
          if (old == 0xFFFFFFFFu) { placed = atomicAdd(&s_used, 1u) < A.max_used; if (!placed) s_full = 1; break; }
          if (old == cl) { placed = true; break; }
          if (*(volatile uint32_t*)&s_full) break;
          slot = (slot + 1) & (uint32_t)(LVB_SLOTS - 1);
        }
        if (placed) atomicAdd(&s_sum[slot], wf);
        else s_full = 1;
      }
    }
    if (!part && self) atomicAdd(&s_self, self);
    __syncthreads();
    if (s_full) {  // (uniform: read after the barrier)
      if (tid == 0) { atomicOr(A.overflow, 1u); A.item_bits[it] = 0; A.item_c[it] = 0x7f7f7f7f; }
      __syncthreads();
      continue;
    }
    unsigned long long const subf = A.rowsub[v], selff = part ? A.rowself[v] : s_self;
    if (r == 0 && tid == 0 && selff) atomicAdd(A.ifix, selff);
    double const sub_d = (double)(long long)subf * A.inv_scale, old_sum = (double)(long long)(selff - subf) * A.inv_scale;
    double const a_old = A.a[cv], kk = A.k[v];
    unsigned long long best = 0;
    int32_t best_c = 0x7f7f7f7f;
    constexpr int LVB_SPT = LVB_SLOTS / LVB_THREADS;
    uint32_t kcl[LVB_SPT];
    double acl[LVB_SPT];
#pragma unroll
    for (int j = 0; j < LVB_SPT; ++j) kcl[j] = s_key[tid + j * LVB_THREADS];
#pragma unroll
    for (int j = 0; j < LVB_SPT; ++j) acl[j] = kcl[j] != 0xFFFFFFFFu ? A.a[kcl[j]] : 0.0;
#pragma unroll
    for (int j = 0; j < LVB_SPT; ++j) {
      uint32_t const cl = kcl[j];
      if (cl == 0xFFFFFFFFu) continue;
      double const sd      = (double)(long long)s_sum[tid + j * LVB_THREADS] * A.inv_scale;
      double const new_sum = (int32_t)cl == cv ? sd - sub_d : sd;
      double const delta   = lv_delta(new_sum, old_sum, acl[j], a_old, kk, A.m, A.resolution);
      unsigned long long const bits = delta > 0.0 ? (unsigned long long)__double_as_longlong(delta) : 0ull;
      if (bits > best || (bits == best && bits && (int32_t)cl < best_c)) { best = bits; best_c = (int32_t)cl; }
    }
    for (int o = 32; o; o >>= 1) {
      unsigned long long const ob = __shfl_xor(best, o);
      int32_t const oc            = __shfl_xor(best_c, o);
      if (ob > best || (ob == best && oc < best_c)) { best = ob; best_c = oc; }
    }
    if (lane == 0) { s_red_bits[wave] = best; s_red_c[wave] = best_c; }
    __syncthreads();
    if (tid == 0) {
      for (int k2 = 1; k2 < LVB_THREADS / 64; ++k2)
        if (s_red_bits[k2] > best || (s_red_bits[k2] == best && s_red_c[k2] < best_c)) { best = s_red_bits[k2]; best_c = s_red_c[k2]; }
      A.item_bits[it] = best;
      A.item_c[it]    = best ? best_c : 0x7f7f7f7f;
      if (best) atomicMax(&A.best_bits[v], best);
    }
    __syncthreads();
  }
}
__global__ void k_lv_big_ties(lv_big_args A)
{
  LV_LOOP(i, (int64_t)A.n_items)
  {
    unsigned long long const bits = A.item_bits[i];
    int32_t const v = A.items[i].x;
    if (bits && bits == A.best_bits[v]) atomicMin(&A.best_c[v], A.item_c[i]);
  }
}
constexpr int LVP_THREADS = 1024, LVP_MAX_R = 8192;  // k_lv_big_partition: one workgroup per big row, a counter per item of the row in LDS
// count[0] += items, count[1] |= "a row needs more than LVP_MAX_R items" (rows of more than 25 M edges: no partition), count[2] += rows, *rescan += d R (the pairs the
// items of the row read when each of them scans the whole row);
// rows[k] = (row, its first item, R, position of its first edge in the big rows' edge list)
__global__ void k_lv_big_items(uint32_t const* off, uint32_t const* pos, int64_t nv, int longer_than, int up_to, int4* items, uint32_t* count, int4* rows, unsigned long long* rescan)
{
  LV_LOOP(v, nv)
  {
    int32_t const d = (int32_t)(off[v + 1] - off[v]);
    if (d > longer_than && d <= up_to) {
      int32_t const R   = (d + LVB_SHARE - 1) / LVB_SHARE;
      uint32_t const at = atomicAdd(count, (uint32_t)R);
      int32_t const b0  = (int32_t)pos[off[v]];
      if (R > LVP_MAX_R) atomicOr(count + 1, 1u);
      atomicAdd(rescan, (unsigned long long)d * (unsigned long long)R);
      rows[atomicAdd(count + 2, 1u)] = make_int4((int32_t)v, (int32_t)at, R, b0);
      for (int32_t r = 0; r < R; ++r) items[at + r] = make_int4((int32_t)v, r, R, b0);
    }
  }
}
// Once per sweep, one workgroup per big row (rows drawn from a queue, longest first): the row's (cluster of destination, weight) pairs grouped by the item whose
// range the cluster falls into -- a counting sort in LDS: count per item, scan, scatter -- so that k_lv_hash_big's items read their own pairs only; the row's
// weight into its own cluster on the way (every item needs it for the gains).  The order inside an item's segment is whatever the atomics give: the sums are
// integers and the choice among equal gains goes by cluster id, so the result does not depend on it.
__global__ void __launch_bounds__(LVP_THREADS) k_lv_big_partition(int4 const* rows, int n_rows, uint32_t const* off, uint32_t const* ecl, unsigned long long const* ewf, int32_t const* c,
                                                                  uint32_t* pcl, unsigned long long* pwf, uint2* seg, unsigned long long* rowself, uint32_t* cursor)
{
  __shared__ uint32_t s_cnt[LVP_MAX_R];
  __shared__ uint32_t s_wsum[LVP_THREADS / 64], s_row;
  __shared__ unsigned long long s_self;
  int const tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (;;) {
    if (tid == 0) s_row = atomicAdd(cursor, 1u);
    __syncthreads();
    int const k = (int)s_row;
    if (k >= n_rows) break;  // (uniform)
    int4 const rw = rows[k];
    int32_t const v = rw.x;
    uint32_t const item0 = (uint32_t)rw.y, R = (uint32_t)rw.z, b0 = (uint32_t)rw.w;
    uint32_t const d = off[v + 1] - off[v];
    int32_t const cv = c[v];
    for (uint32_t i = tid; i < R; i += LVP_THREADS) s_cnt[i] = 0;
    if (tid == 0) s_self = 0;
    __syncthreads();
    unsigned long long self = 0;
    constexpr int U = 8;  // loads in flight per thread: the longest rows (10^7 edges at the coarse levels of RMAT-26) are walked by ONE workgroup
    for (uint32_t i0 = tid; i0 < d; i0 += LVP_THREADS * U) {
      uint32_t cls[U];
#pragma unroll
      for (int j = 0; j < U; ++j) { uint32_t const i = i0 + (uint32_t)j * LVP_THREADS; cls[j] = i < d ? ecl[b0 + i] : 0xFFFFFFFFu; }
#pragma unroll
      for (int j = 0; j < U; ++j) {
        uint32_t const i = i0 + (uint32_t)j * LVP_THREADS;
        if (i < d) {
          atomicAdd(&s_cnt[lvb_range(cls[j], R)], 1u);
          if ((int32_t)cls[j] == cv) self += ewf[b0 + i];
        }
      }
    }
    for (int o = 32; o; o >>= 1) self += __shfl_xor(self, o);
    if (lane == 0 && self) atomicAdd(&s_self, self);
    __syncthreads();
    // exclusive scan of the R counts: a thread takes a run of consecutive counters
    uint32_t const per = (R + LVP_THREADS - 1) / LVP_THREADS, lo = (uint32_t)tid * per, hi = lo + per < R ? lo + per : R;
    uint32_t sum = 0;
    for (uint32_t j = lo; j < hi; ++j) sum += s_cnt[j];
    uint32_t inc = sum;
    for (int o = 1; o < 64; o <<= 1) {
      uint32_t const t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += s_wsum[w];
    uint32_t run = before + inc - sum;
    for (uint32_t j = lo; j < hi; ++j) {
      uint32_t const n = s_cnt[j];
      s_cnt[j]         = run;
      seg[item0 + j]   = make_uint2(b0 + run, n);
      run += n;
    }
    __syncthreads();
    for (uint32_t i0 = tid; i0 < d; i0 += LVP_THREADS * U) {
      uint32_t cls[U];
      unsigned long long wfs[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        uint32_t const i = i0 + (uint32_t)j * LVP_THREADS;
        cls[j] = 0; wfs[j] = 0;
        if (i < d) { cls[j] = ecl[b0 + i]; wfs[j] = ewf[b0 + i]; }
      }
#pragma unroll
      for (int j = 0; j < U; ++j) {
        if (i0 + (uint32_t)j * LVP_THREADS < d) {
          uint32_t const at = b0 + atomicAdd(&s_cnt[lvb_range(cls[j], R)], 1u);
          pcl[at] = cls[j];
          pwf[at] = wfs[j];
        }
      }
    }
    if (tid == 0) rowself[v] = s_self;
    __syncthreads();
  }
}
// once per level: fixed-point weights of the big rows' edges, self-loop weight per row
__global__ void k_lv_big_prep(int32_t const* hs, int32_t const* hd, double const* hw, int64_t n, double scale, unsigned long long* ewf, unsigned long long* rowsub)
{
  LV_LOOP(q, n)
  {
    unsigned long long const wf = (unsigned long long)__double2ll_rn(hw[q] * scale);
    ewf[q] = wf;
    if (hs[q] == hd[q]) atomicAdd(&rowsub[hs[q]], wf);
  }
}
// once per sweep: the cluster of every big-row edge's destination (the items of a row then re-read 12 sequential bytes per edge)
__global__ void k_lv_big_gather(int32_t const* hd, int32_t const* c, int64_t n, uint32_t* ecl) { LV_LOOP(q, n) ecl[q] = (uint32_t)c[hd[q]]; }
__global__ void k_lv_mid_rows(uint32_t const* off, int64_t nv, int lo, int hi, int32_t* rows, uint32_t* count)
{
  LV_LOOP(v, nv)
  {
    int32_t const d = (int32_t)(off[v + 1] - off[v]);
    if (d > lo && d <= hi) rows[atomicAdd(count, 1u)] = (int32_t)v;
  }
}
// OPT-IN (CUGRAPH_AMD_LOUVAIN_HUB=hash; the default keeps the sorted path for hubs, see run_level): hub rows (more than LVH_B
// edges -- 36 % of the edges of RMAT-22, 56 % of RMAT-26) WITHOUT sorting either: their edges are
// compacted once per level (hub list, grouped by row); row v with deg edges owns the table region [2 * first, 2 * first + 2 * deg)
// of a global open-addressing table (first = position of the row's first edge in the hub list), keyed by the destination's cluster
// inside the region.  k_lv_hub_insert adds the fixed-point weights (runs of equal (row, cluster) on neighbouring lanes are summed
// in the wavefront first: a hub whose neighbours sit in one giant cluster would otherwise serialise its atomics on one address),
// k_lv_hub_eval<0/1> walks the table slots flat -- slot t belongs to the row of hub edge t / 2 -- and reduces the best gain / the
// smallest cluster among the best per row exactly as the sorted path's k_segment_best does.  Same integers, same gains, same ties.
struct lv_hub_args {
  int32_t const* hs; int32_t const* hd; double const* hw; uint32_t const* hrow0; uint32_t const* off;  // hub list + first hub position of the edge's row
  int32_t const* c; double const* k; double const* a; double m, resolution, scale, inv_scale; int64_t n_h;
  unsigned long long* keys; unsigned long long* sums;  // [2 * n_h]: keys = cluster id, ~0 = empty (set by the caller), sums = 0
  unsigned long long* selffix; unsigned long long* subfix; unsigned long long* best_bits; int32_t* best_c;  // [nv]
};
__global__ void k_lv_hub_insert(lv_hub_args A)
{
  int const lane       = threadIdx.x & 63;
  int64_t const wave   = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t const nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t q0 = wave * 64; q0 < A.n_h; q0 += nwaves * 64) {
    int64_t const q  = q0 + lane;
    bool const valid = q < A.n_h;
    int32_t v = -1, cl = -1, cv = -2;
    long long wf = 0;
    uint32_t r0 = 0, size = 2;
    if (valid) {
      v            = A.hs[q];
      int32_t const u = A.hd[q];
      cl           = A.c[u];
      cv           = A.c[v];
      wf           = __double2ll_rn(A.hw[q] * A.scale);
      r0           = A.hrow0[q];
      size         = 2u * (uint32_t)(A.off[v + 1] - A.off[v]);
      if (u == v) atomicAdd(&A.subfix[v], (unsigned long long)wf);  // self-loop
    }
    // runs of equal (row, cluster) on neighbouring lanes: one table update per run
    int32_t const vp = __shfl_up(v, 1), cp = __shfl_up(cl, 1), vn = __shfl_down(v, 1), cn = __shfl_down(cl, 1);
    bool const head = lane == 0 || v != vp || cl != cp;
    bool const end  = lane == 63 || v != vn || cl != cn;
    bool open;
    long long const sum = seg_scan64(wf, head, lane, open);
    if (valid && end) {
      uint64_t const base = 2ull * (uint64_t)r0;
      uint32_t i          = (uint32_t)(((uint64_t)((uint32_t)cl * 0x9E3779B1u) * (uint64_t)size) >> 32);
      unsigned long long const key = (unsigned long long)(uint32_t)cl;
      for (;;) {
        unsigned long long const old = atomicCAS(&A.keys[base + i], ~0ull, key);
        if (old == ~0ull || old == key) break;
        if (++i == size) i = 0;
      }
      atomicAdd(&A.sums[base + i], (unsigned long long)sum);
      if (cl == cv) atomicAdd(&A.selffix[v], (unsigned long long)sum);
    }
  }
}
template <int PHASE>
__global__ void k_lv_hub_eval(lv_hub_args A)
{
  int const lane       = threadIdx.x & 63;
  int64_t const wave   = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t const nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int64_t const nslots = 2 * A.n_h;
  for (int64_t t0 = wave * 64; t0 < nslots; t0 += nwaves * 64) {
    int64_t const t  = t0 + lane;
    bool const valid = t < nslots;
    unsigned long long const key = valid ? A.keys[t] : ~0ull;
    int32_t const v = valid ? A.hs[t >> 1] : -1;  // the slot lies in the region of this row
    unsigned long long bits = 0;
    int32_t const cl = (int32_t)(uint32_t)key;
    if (key != ~0ull) {
      int32_t const cv     = A.c[v];
      double const s       = (double)(long long)A.sums[t] * A.inv_scale;
      double const sub     = (double)(long long)A.subfix[v] * A.inv_scale;
      double const old_sum = (double)(long long)(A.selffix[v] - A.subfix[v]) * A.inv_scale;
      double const new_sum = cl == cv ? s - sub : s;
      double const delta   = lv_delta(new_sum, old_sum, A.a[cl], A.a[cv], A.k[v], A.m, A.resolution);
      if (delta > 0.0) bits = (unsigned long long)__double_as_longlong(delta);
    }
    if (PHASE == 0) {  // maximum over the lanes of one row (consecutive lanes), then one atomic per (wavefront, row)
      int32_t const vprev = __shfl_up(v, 1), vnext = __shfl_down(v, 1);
      bool vhead = lane == 0 || v != vprev;
      unsigned long long mx = bits;
      unsigned f = vhead ? 1u : 0u;
      for (int o = 1; o < 64; o <<= 1) {
        unsigned long long const tm = __shfl_up(mx, o);
        unsigned const tf           = __shfl_up(f, o);
        if (lane >= o && !f) { mx = tm > mx ? tm : mx; f |= tf; }
      }
      bool const vend = lane == 63 || v != vnext;
      if (valid && vend && mx) atomicMax(&A.best_bits[v], mx);
    } else {
      if (bits && bits == A.best_bits[v]) atomicMin(&A.best_c[v], cl);
    }
  }
}
// hub rows (more than LVH_B edges): their edges, compacted once per level, go through the sorted path
__global__ void k_lv_hub_flags(int32_t const* src, uint32_t const* off, int64_t ne, int32_t longer_than, uint32_t* flag)
{
  LV_LOOP(e, ne) { int32_t const v = src[e]; flag[e] = ((int32_t)(off[v + 1] - off[v]) > longer_than) ? 1u : 0u; }
}
__global__ void k_lv_hub_compact(int32_t const* src, int32_t const* dst, double const* w, uint32_t const* off, uint32_t const* flag, uint32_t const* pos, int64_t ne,
                                 int32_t* hs, int32_t* hd, double* hw, uint32_t* hrow0)
{
  LV_LOOP(e, ne) if (flag[e]) { uint32_t const q = pos[e]; hs[q] = src[e]; hd[q] = dst[e]; hw[q] = w[e]; hrow0[q] = pos[off[src[e]]]; }
}
__global__ void k_best_finalize(unsigned long long const* best_bits, int32_t* best_c, double* best_d, int64_t nv)
{
  LV_LOOP(v, nv)
  {
    unsigned long long const b = best_bits[v];
    best_d[v] = __longlong_as_double((long long)b);
    if (!b) best_c[v] = -1;
  }
}

// moves wanted in each direction: count[0] = "down" (best_c < c), count[1] = "up".  Few large workgroups, one pair of atomics per
// workgroup: with one atomic per wavefront the 65 Ki wavefronts of a 4 M-vertex level queued up on one cache line (0.3-1.3 ms for a
// kernel that reads 80 MB; the same finding as for the BFS counters)
__global__ void __launch_bounds__(1024) k_count_moves(int32_t const* c, int32_t const* best_c, double const* best_d, double min_gain, int64_t nv, uint32_t* count)
{
  __shared__ uint32_t s_cnt[2];
  if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  uint32_t up = 0, down = 0;
  LV_LOOP(v, nv)
  {
    bool const want = best_d[v] > min_gain;
    bool const u    = best_c[v] > c[v];
    up += want && u;
    down += want && !u;
  }
  for (int o = 32; o > 0; o >>= 1) { up += __shfl_xor(up, o); down += __shfl_xor(down, o); }
  if ((threadIdx.x & 63) == 0) { if (up) atomicAdd(&s_cnt[1], up); if (down) atomicAdd(&s_cnt[0], down); }
  __syncthreads();
  if (threadIdx.x < 2 && s_cnt[threadIdx.x]) atomicAdd(count + threadIdx.x, s_cnt[threadIdx.x]);
}
// the move, and the two cluster weights it changes (fixed point: exact, order-free).  The changes of a workgroup's 1024 vertices are first added up per
// cluster in an LDS table and leave as ONE global atomic per (workgroup, cluster): at a contracted level most moves go into a handful of giant clusters,
// and one device atomic per vertex on those few addresses serialised (0.5 ms for 2.5 M vertices at the second level of RMAT-22; round 6)
constexpr int LVA_THREADS = 1024, LVA_SLOTS = 4096;  // at most 2 * LVA_THREADS keys per round: the table stays half empty
__global__ void __launch_bounds__(LVA_THREADS) k_apply_moves(int32_t* c, int32_t const* best_c, double const* best_d, double min_gain, int up_down, int64_t nv,
                                                             long long const* kfix, unsigned long long* afix)
{
  __shared__ uint32_t s_key[LVA_SLOTS];
  __shared__ unsigned long long s_val[LVA_SLOTS];
  for (int i = threadIdx.x; i < LVA_SLOTS; i += LVA_THREADS) { s_key[i] = 0xFFFFFFFFu; s_val[i] = 0; }
  __syncthreads();
  auto add = [&](uint32_t cl, unsigned long long x) {
    uint32_t slot = ((cl * 0x9E3779B1u) ^ (cl >> 15)) & (uint32_t)(LVA_SLOTS - 1);
    for (;;) {
      uint32_t const old = atomicCAS(&s_key[slot], 0xFFFFFFFFu, cl);
      if (old == 0xFFFFFFFFu || old == cl) break;
      slot = (slot + 1) & (uint32_t)(LVA_SLOTS - 1);
    }
    atomicAdd(&s_val[slot], x);
  };
  int64_t const rounds = (nv + LVA_THREADS - 1) / LVA_THREADS;
  for (int64_t r = blockIdx.x; r < rounds; r += gridDim.x) {  // (uniform trip count: the barriers below are reached by every thread)
    int64_t const v = r * LVA_THREADS + threadIdx.x;
    if (v < nv && best_d[v] > min_gain && ((best_c[v] > c[v]) == (up_down != 0))) {
      unsigned long long const kv = (unsigned long long)kfix[v];
      add((uint32_t)best_c[v], kv);
      add((uint32_t)c[v], 0ull - kv);
      c[v] = best_c[v];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LVA_SLOTS; i += LVA_THREADS) {
      uint32_t const cl = s_key[i];
      if (cl != 0xFFFFFFFFu) {
        unsigned long long const x = s_val[i];
        if (x) atomicAdd(&afix[cl], x);
        s_key[i] = 0xFFFFFFFFu; s_val[i] = 0;
      }
    }
    __syncthreads();
  }
}
__global__ void k_fix_to_double(unsigned long long const* fix, int64_t n, double inv_scale, double* out)
{
  LV_LOOP(i, n) out[i] = (double)(long long)fix[i] * inv_scale;
}

// fixed-order reductions: every workgroup folds one 64 Ki-element chunk (thread-strided, then a tree), one workgroup folds the
// chunk sums in index order -- the result does not depend on the launch geometry
constexpr int64_t LV_RCHUNK = 65536;
__device__ __forceinline__ void lv_block_fold(double s, double* out)
{
  __shared__ double red[1024];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) *out = red[0];
}
// sum of an array of fixed-point values into *out (the sorted / global-table paths leave the rows' own-cluster weights in selffix[])
__global__ void __launch_bounds__(1024) k_lv_sum_u64(unsigned long long const* x, int64_t n, unsigned long long* out)
{
  __shared__ unsigned long long s_acc;
  if (threadIdx.x == 0) s_acc = 0;
  __syncthreads();
  unsigned long long s = 0;
  for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 1024) s += x[i];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0 && s) atomicAdd(&s_acc, s);
  __syncthreads();
  if (threadIdx.x == 0 && s_acc) atomicAdd(out, s_acc);
}
__global__ void __launch_bounds__(1024) k_part_squares(double const* a, int64_t n, double* part)
{
  int64_t const b = (int64_t)blockIdx.x * LV_RCHUNK, e = b + LV_RCHUNK < n ? b + LV_RCHUNK : n;
  double s = 0.0;
  for (int64_t i = b + threadIdx.x; i < e; i += 1024) s += a[i] * a[i];
  lv_block_fold(s, part + blockIdx.x);
}
__global__ void __launch_bounds__(1024) k_part_sum(double const* w, int64_t n, double* part)
{
  int64_t const b = (int64_t)blockIdx.x * LV_RCHUNK, e = b + LV_RCHUNK < n ? b + LV_RCHUNK : n;
  double s = 0.0;
  for (int64_t i = b + threadIdx.x; i < e; i += 1024) s += w[i];
  lv_block_fold(s, part + blockIdx.x);
}
__global__ void __launch_bounds__(1024) k_fold_parts(double const* part, int64_t n, double* out)
{
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) s += part[i];
  lv_block_fold(s, out);
}

__global__ void k_mark_labels(int32_t const* c, int64_t nv, uint32_t* used) { LV_LOOP(v, nv) used[c[v]] = 1u; }
__global__ void k_relabel(int32_t* c, uint32_t const* rank, int64_t nv) { LV_LOOP(v, nv) c[v] = (int32_t)rank[c[v]]; }
__global__ void k_compose(int32_t* part, int32_t const* c, int64_t n) { LV_LOOP(i, n) part[i] = c[part[i]]; }
__global__ void k_copy_i32(int32_t* dst, int32_t const* src, int64_t n) { LV_LOOP(i, n) dst[i] = src[i]; }

// contraction: edges sorted by (cluster of src, cluster of dst); segment number = pos[i] (exclusive scan of the heads).  The head
// of a segment emits the coarse edge's endpoints; the weights are summed in fixed point: a wavefront reduces its 64 sorted
// entries by segment and adds one value per (wavefront, segment) with an integer atomic
// (wfix_in != nullptr: the entries carry fixed-point weights already -- the second stage of the partitioned contraction, which adds up
// what the ranks aggregated locally; the total is the same integer as the single-GPU sum)
__global__ void k_coarse_edges(uint64_t const* keys, uint32_t const* perm, uint32_t const* head, uint32_t const* pos, double const* w, int64_t ne,
                               int shift,
                               double scale, int32_t* csrc, int32_t* cdst, unsigned long long* cwfix, unsigned long long const* wfix_in = nullptr)
{
  int const lane       = threadIdx.x & 63;
  int64_t const wave   = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t const nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i0 = wave * 64; i0 < ne; i0 += nwaves * 64) {
    int64_t const i  = i0 + lane;
    bool const valid = i < ne;
    uint32_t seg = 0;
    long long wf = 0;
    bool hd = true;
    if (valid) {
      hd  = head[i] != 0;
      seg = pos[i] - (hd ? 0u : 1u);  // pos = exclusive scan of the heads: a segment's later entries already count their own head
      wf  = wfix_in ? (long long)wfix_in[perm[i]] : __double2ll_rn(w[perm[i]] * scale);
      if (hd) { csrc[seg] = (int32_t)(keys[i] >> shift); cdst[seg] = (int32_t)(keys[i] & ((1ull << shift) - 1ull)); }
    }
    bool open;
    long long const sum = seg_scan64(wf, hd || lane == 0, lane, open);
    bool const end = valid && (i + 1 >= ne || lane == 63 || head[i + 1] != 0);
    if (end) atomicAdd(&cwfix[seg], (unsigned long long)sum);
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Partitioned Louvain (cugraph_louvain on a graph from cugraph_graph_create_mg; BASELINE config 5).  Replaces the multi_gpu = true
// path of detail::louvain (cpp/src/community/louvain_impl.cuh:78-262): update_clustering_by_delta_modularity's collect_values_for_keys /
// host_scalar_allreduce (detail/common_methods.cuh:200, 259-447) and the MG coarsen_graph (cpp/src/structure/coarsen_graph_impl.cuh).
// A level's vertices v = 0 .. nv - 1 are dealt cyclically (owner v % P); a rank holds the out-edges of its vertices (sorted by source,
// destination) and FULL copies of the per-vertex / per-cluster vectors (labels c, vertex weights k, cluster weights a: 20 bytes per vertex).
// A sweep runs the single-GPU kernels on the local edges -- every per-vertex quantity they produce depends only on that vertex's edges and
// the full vectors -- then the owners' new labels are merged into every rank's c (one push of nv / P labels per rank), and the cluster
// weights are rebuilt from c and k with integer atomics.  All sums that decide a move are 64-bit fixed point, so the partitioned run takes
// the same moves as the single-GPU run, vertex for vertex, whatever the number of ranks.  The modularity of a sweep = (sum over ranks, in rank
// order, of the local intra-cluster weight) / m - resolution * sum a^2 / m^2 with the second sum formed by every rank over the full vector in
// the single-GPU chunk order: bit-equal to the single-GPU value whenever the edge weights are integers (any fp64 sum of them is exact).
// Contraction: local (cluster, cluster) aggregation in fixed point, one all-to-all of the coarse edges to the owner of the source cluster,
// a second integer aggregation there.
struct lv_mg_t {
  comm_t* c{nullptr};
  handle_t const* h{nullptr};
  int P{1}, rank{0}, channel{1};
  comm_window_t* stage{nullptr};  // [P][Lc] 8-byte slots: the owners' values of one merge
  int64_t Lc{0};
  void level_begin(int64_t nv)
  {
    Lc    = std::max<int64_t>((nv + P - 1) / P, 1);
    stage = c->window_create((size_t)P * (size_t)Lc * 8);
  }
  void level_end()
  {
    h->sync();
    if (stage) c->window_free(stage);
    stage = nullptr;
  }
  double sum_f64(double x)
  {
    std::vector<double> all(P);
    c->host_allgather(&x, sizeof(x), all.data());
    double s = 0.0;
    for (double y : all) s += y;  // rank order
    return s;
  }
  unsigned long long sum_u64(unsigned long long x)
  {
    std::vector<unsigned long long> all(P);
    c->host_allgather(&x, sizeof(x), all.data());
    unsigned long long t = 0;
    for (auto y : all) t += y;
    return t;
  }
  void sum_u32x2(uint32_t* v)
  {
    uint64_t mine[2] = {v[0], v[1]};
    std::vector<uint64_t> all((size_t)2 * P);
    c->host_allgather(mine, sizeof(mine), all.data());
    uint64_t a = 0, b = 0;
    for (int r = 0; r < P; ++r) { a += all[2 * r]; b += all[2 * r + 1]; }
    v[0] = (uint32_t)std::min<uint64_t>(a, 0xffffffffu); v[1] = (uint32_t)std::min<uint64_t>(b, 0xffffffffu);
  }
  template <typename T> void merge_owned(T* full, int64_t nv);  // full[v] <- the value rank v % P holds, on every rank
};

template <typename T>
__global__ void k_take_owned(T const* full, int64_t nv, int P, int rank, T* out)
{
  LV_LOOP(i, (nv - rank + P - 1) / P) out[i] = full[i * P + rank];
}
template <typename T>
__global__ void k_interleave(T const* stage, int64_t nv, int P, int64_t Lc, T* full)
{
  LV_LOOP(v, nv) full[v] = stage[(v % P) * Lc + v / P];
}
template <typename T>
void lv_mg_t::merge_owned(T* full, int64_t nv)
{
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "4- or 8-byte values");
  int64_t const mine = nv > rank ? (nv - rank + P - 1) / P : 0;
  dvec<T> own((size_t)std::max<int64_t>(mine, 1));
  if (mine > 0) hipLaunchKernelGGL(k_take_owned<T>, grid_for(mine, kBlock, 8192), kBlock, 0, h->stream, (T const*)full, nv, P, rank, own.data());
  comm_push_desc_t d{};
  for (int r = 0; r < P; ++r) { d.dst[r] = static_cast<T*>(stage->peer[r]) + (int64_t)rank * Lc; d.src[r] = own.data(); d.words[r] = mine * (int64_t)(sizeof(T) / 4); }
  d.n = P;
  c->push_multi(h->stream, d);
  c->wait(h->stream, channel, c->signal(h->stream, channel));
  if (nv > 0) hipLaunchKernelGGL(k_interleave<T>, grid_for(nv, kBlock, 8192), kBlock, 0, h->stream, (T const*)stage->local, nv, P, Lc, full);
  // the next merge overwrites the staging rows: every rank must have consumed them first
  c->wait(h->stream, channel, c->signal(h->stream, channel));
  h->sync();  // (`own` is released here)
}

__global__ void k_cluster_weights(int32_t const* c, long long const* kfix, int64_t nv, unsigned long long* afix)
{
  LV_LOOP(v, nv) { unsigned long long const kv = (unsigned long long)kfix[v]; if (kv) atomicAdd(&afix[c[v]], kv); }
}

struct level_t {
  int64_t nv{0}, ne{0};
  dvec<int32_t> src, dst;
  dvec<uint32_t> off;  // edge positions are unsigned 32-bit words (a level may hold 2^31 or more edges)
  dvec<double> w;
  bool have_off{false};  // off[] holds the rows' offsets
};

// row offsets of a level whose edges are stored grouped by source, ascending (every level is: CSR order at level 0, the contraction's segment
// order afterwards): every row's first and last position are found at the boundaries of the source column (one streaming pass), the row lengths
// are scanned (round 6: was a histogram of the sources -- a partition by the top bits + windowed LDS counters, 1 ms per level at RMAT-22)
__global__ void k_row_bounds(int32_t const* src, int64_t ne, uint32_t* first, uint32_t* last)  // both zero on entry
{
  LV_LOOP(i, ne)
  {
    int32_t const v = src[i];
    if (i == 0 || src[i - 1] != v) first[v] = (uint32_t)i;
    if (i + 1 == ne || src[i + 1] != v) last[v] = (uint32_t)(i + 1);
  }
}
__global__ void k_row_lengths(uint32_t const* first, uint32_t* last, int64_t nv) { LV_LOOP(v, nv) last[v] -= first[v]; }
void build_offsets(handle_t const& h, level_t& L)
{
  if (L.have_off) return;  // (level 0 of a single-GPU graph: the graph's own offsets)
  L.off.resize_discard((size_t)L.nv + 1);
  dvec<uint32_t> fl(2 * ((size_t)L.nv + 1));
  uint32_t* const first = fl.data();
  uint32_t* const last  = fl.data() + L.nv + 1;
  HIP_TRY(hipMemsetAsync(fl.data(), 0, 2 * ((size_t)L.nv + 1) * sizeof(uint32_t), h.stream));
  if (L.ne > 0) {
    hipLaunchKernelGGL(k_row_bounds, grid_for(L.ne, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)L.src.data(), L.ne, first, last);
    hipLaunchKernelGGL(k_row_lengths, grid_for(L.nv, kBlock, 8192), kBlock, 0, h.stream, (uint32_t const*)first, last, L.nv);
  }
  exclusive_scan_u32(h, last, L.off.data(), L.nv + 1);
  h.sync();  // `fl` is released here
  L.have_off = true;
}

void sort_pairs(handle_t const& h, dvec<uint64_t>& keys, dvec<uint32_t>& vals, int64_t n, int bits)
{
  if (n <= 1) return;
  dvec<uint64_t> kt((size_t)n);
  dvec<uint32_t> vt((size_t)n);
  radix_sort_u64_u32(h, keys.data(), vals.data(), kt.data(), vt.data(), n, 0, bits);
}

// ---- numbering of a contracted level.  graph_contraction (detail/common_methods.cuh:231-263) builds the coarse graph with coarsen_graph(...,
// renumber = true), i.e. the coarse vertices are numbered as every graph creation numbers vertices: by degree, descending
// (structure/renumber_edgelist_impl.cuh, "4. sort local vertices by degree (descending)": a stable key sort over the id-sorted vertex list, so equal
// degrees keep ascending label order); degree = coarse edges leaving the vertex = its distinct neighbour clusters.  The ids are not cosmetic: the next
// level breaks equal gains towards the smaller cluster id (reduce_op_t) and moves up or down by comparing ids.  Rounds 1-5 numbered the coarse vertices
// by label order and missed the reference's karate goldens at resolution 1 (cpp/tests/community/louvain_test.cpp:228-237: 0.39907956, three levels).
__global__ void k_lv_row_lengths_of(uint32_t const* off, int64_t n, uint32_t* len) { LV_LOOP(v, n) len[v] = (uint32_t)off[v + 1] - (uint32_t)off[v]; }
__global__ void k_lv_degree_keys(uint32_t const* deg, int64_t n, uint64_t* keys, uint32_t* vals)
{
  LV_LOOP(v, n) { keys[v] = (uint64_t)(0xFFFFFFFFu - deg[v]); vals[v] = (uint32_t)v; }
}
__global__ void k_lv_new_ids(uint64_t const* keys, uint32_t const* old_of, int64_t n, uint32_t* new_id, uint32_t* len)
{
  LV_LOOP(r, n) { new_id[old_of[r]] = (uint32_t)r; len[r] = 0xFFFFFFFFu - (uint32_t)keys[r]; }
}
// deg[ncl] (coarse out-degrees in the label-order ids) -> new_id[old] (position in the order by (degree descending, old id ascending)) and
// new_off[ncl + 1] (the rows' offsets in that order)
void coarse_degree_order(handle_t const& h, uint32_t const* deg, int64_t ncl, dvec<uint32_t>& new_id, dvec<uint32_t>& new_off)
{
  size_t const n1 = (size_t)std::max<int64_t>(ncl, 1);
  dvec<uint64_t> keys(n1);
  dvec<uint32_t> old_of(n1), len(n1 + 1);
  new_id.resize_discard(n1 + 1);
  new_off.resize_discard(n1 + 1);
  HIP_TRY(hipMemsetAsync(len.data(), 0, (n1 + 1) * sizeof(uint32_t), h.stream));
  if (ncl > 0) {
    hipLaunchKernelGGL(k_lv_degree_keys, grid_for(ncl, kBlock, 8192), kBlock, 0, h.stream, deg, ncl, keys.data(), old_of.data());
    sort_pairs(h, keys, old_of, ncl, 32);  // LSD radix passes are stable: equal degrees stay in ascending id order
    hipLaunchKernelGGL(k_lv_new_ids, grid_for(ncl, kBlock, 8192), kBlock, 0, h.stream, (uint64_t const*)keys.data(), (uint32_t const*)old_of.data(), ncl, new_id.data(), len.data());
  }
  exclusive_scan_u32(h, len.data(), new_off.data(), ncl + 1);
  h.sync();  // (temporaries die here)
}
// the rows of a coarse edge list (sorted by source in the OLD ids, offsets old_off) move to where the new numbering puts them; a row keeps its internal order
__global__ void k_lv_renumber_rows(int32_t const* src, int32_t const* dst, unsigned long long const* wfix, int64_t ne, uint32_t const* old_off, uint32_t const* new_off,
                                   uint32_t const* new_id, double inv_scale, int32_t* src2, int32_t* dst2, double* w2)
{
  LV_LOOP(p, ne)
  {
    int32_t const o  = src[p];
    uint32_t const r = new_id[o];
    uint32_t const q = new_off[r] + ((uint32_t)p - (uint32_t)old_off[o]);
    src2[q] = (int32_t)r;
    dst2[q] = (int32_t)new_id[dst[p]];
    w2[q]   = (double)(long long)wfix[p] * inv_scale;
  }
}

// fixed-order sum of n values produced chunk-wise by `launch_parts(parts)`
template <typename F>
void chunked_sum(handle_t const& h, int64_t n, dvec<double>& parts, double* out, F&& launch_parts)
{
  int64_t const np = std::max<int64_t>(1, (n + LV_RCHUNK - 1) / LV_RCHUNK);
  if (parts.size() < (size_t)np) parts.resize_discard((size_t)np);
  if (n > 0) launch_parts((int)np, parts.data());
  else HIP_TRY(hipMemsetAsync(parts.data(), 0, sizeof(double), h.stream));
  hipLaunchKernelGGL(k_fold_parts, 1, 1024, 0, h.stream, (double const*)parts.data(), np, out);
}

// fixed-point scale 2^s for sums bounded by the total edge weight m: |sum| * 2^s < 2^61
double fixed_scale(double m)
{
  int e = 0;
  (void)std::frexp(m > 1.0 ? m : 1.0, &e);  // m < 2^e
  return std::ldexp(1.0, 61 - e);
}

struct louvain_stats_t { int sweeps{0}, sweeps_in_level{0}; };

// one level (the body of the while loop of detail::louvain, louvain_impl.cuh:78-262): accepted clustering and its modularity
double run_level(handle_t const& h, level_t const& L, double m, double threshold, double resolution, double noise_floor, dvec<int32_t>& accepted,
                 louvain_stats_t& st, lv_mg_t* mg = nullptr)
{
  int64_t const nv = L.nv, ne = L.ne;
  int const g_v = grid_for(nv, kBlock, 8192), g_e = grid_for(ne, kBlock, 8192);
  double const scale = fixed_scale(m);
  dvec<double> k((size_t)nv), a((size_t)nv), best_d((size_t)nv), scal(2), parts;
  dvec<long long> kfix((size_t)nv);
  dvec<unsigned long long> afix((size_t)nv), vfix((size_t)nv * 3), segfix;
  dvec<int32_t> c((size_t)nv), best_c((size_t)nv);
  // rows of at most LVH_B edges: hash path (k_lv_hash_chunks); longer rows ("hubs"): their edges are compacted once per level
  // and take the sorted path.  CUGRAPH_AMD_LOUVAIN_HASH=0: every row takes the sorted path (round 2's behaviour).
  bool const use_hash = ne > 0 && !(getenv("CUGRAPH_AMD_LOUVAIN_HASH") && atoi(getenv("CUGRAPH_AMD_LOUVAIN_HASH")) == 0);
  level_t Lh;  // the hub rows' edges (use_hash) -- src / dst / w only
  dvec<uint32_t> hrow0;  // position in the hub list of the first edge of every hub edge's row
  int64_t n_sorted = ne;
  // rows of LVH_B < degree <= LVM_MAX edges: one workgroup per row (k_lv_hash_rows; two table sizes).  CUGRAPH_AMD_LOUVAIN_MID=0: they
  // stay with the hubs (round 3's behaviour)
  bool const use_mid = use_hash && !(getenv("CUGRAPH_AMD_LOUVAIN_MID") && atoi(getenv("CUGRAPH_AMD_LOUVAIN_MID")) == 0);
  // rows of more than LVM_MAX edges: LDS tables too, several work items per row (k_lv_hash_big); CUGRAPH_AMD_LOUVAIN_BIG=0: sorted path
  bool const use_big = use_mid && !(getenv("CUGRAPH_AMD_LOUVAIN_BIG") && atoi(getenv("CUGRAPH_AMD_LOUVAIN_BIG")) == 0);
  dvec<int32_t> mid_rows[2];
  dvec<uint32_t> mid_count(5), big_cursor(2);  // big_cursor: [0] k_lv_hash_big's items, [1] k_lv_big_partition's rows
  uint32_t n_mid[5] = {0, 0, 0, 0, 0};  // [2] = work items of the big rows, [3] != 0: a row needs more items than k_lv_big_partition counts in LDS, [4] = big rows
  dvec<int4> big_items, big_rows;
  dvec<uint32_t> big_pcl;
  dvec<unsigned long long> big_pwf, big_rowself;
  dvec<uint2> big_seg;
  dvec<unsigned long long> big_rescan(1);
  unsigned long long big_rescan_h = 0;  // sum over the big rows of d R
  dvec<unsigned long long> big_bits, big_ewf, big_rowsub;
  dvec<uint32_t> big_ecl;
  dvec<int32_t> big_c;
  dvec<long long> chunk_range;
  dvec<uint16_t> chunk_rs;
  dvec<int4> ca;
  dvec<unsigned long long> chunk_int;
  if (use_hash) {
    ca.resize_discard((size_t)std::max<int64_t>(nv, 1));
    chunk_int.resize_discard((size_t)((ne + LVH_B - 1) / LVH_B));
    chunk_range.resize_discard((size_t)((ne + LVH_B - 1) / LVH_B) * 2);
    chunk_rs.resize_discard((size_t)ne);
    hipLaunchKernelGGL(k_lv_chunk_prep, g_e, kBlock, 0, h.stream, (int32_t const*)L.src.data(), (int32_t const*)L.dst.data(), (uint32_t const*)L.off.data(), ne, chunk_range.data(),
                       chunk_rs.data());
    dvec<uint32_t> flag((size_t)ne + 1), pos((size_t)ne + 1);
    hipLaunchKernelGGL(k_lv_hub_flags, g_e, kBlock, 0, h.stream, (int32_t const*)L.src.data(), (uint32_t const*)L.off.data(), ne, (int32_t)(use_mid ? LVM_MAX : LVH_B),
                       flag.data());
    HIP_TRY(hipMemsetAsync(flag.data() + ne, 0, sizeof(uint32_t), h.stream));
    exclusive_scan_u32(h, flag.data(), pos.data(), ne + 1);
    if (use_mid) {
      size_t const cap = (size_t)(ne / LVH_B + 2);  // rows of more than LVH_B edges
      mid_rows[0].resize_discard(cap); mid_rows[1].resize_discard(cap);
      HIP_TRY(hipMemsetAsync(mid_count.data(), 0, 5 * sizeof(uint32_t), h.stream));
      hipLaunchKernelGGL(k_lv_mid_rows, g_v, kBlock, 0, h.stream, (uint32_t const*)L.off.data(), nv, LVH_B, LVM_MAX / 2, mid_rows[0].data(), mid_count.data());
      hipLaunchKernelGGL(k_lv_mid_rows, g_v, kBlock, 0, h.stream, (uint32_t const*)L.off.data(), nv, LVM_MAX / 2, LVM_MAX, mid_rows[1].data(), mid_count.data() + 1);
      if (use_big) {
        big_items.resize_discard((size_t)(ne / LVB_SHARE + ne / LVM_MAX + 2));  // sum over the big rows of ceil(degree / LVB_SHARE)
        // the items of the longest rows first (two classes; inside a class in whatever order the atomics land): k_lv_hash_big draws them in this order
        constexpr int kLongRow = 16 * LVB_SHARE;
        big_rows.resize_discard((size_t)(ne / LVM_MAX + 2));
        HIP_TRY(hipMemsetAsync(big_rescan.data(), 0, sizeof(unsigned long long), h.stream));
        hipLaunchKernelGGL(k_lv_big_items, g_v, kBlock, 0, h.stream, (uint32_t const*)L.off.data(), (uint32_t const*)pos.data(), nv, kLongRow, INT32_MAX, big_items.data(),
                           mid_count.data() + 2, big_rows.data(), big_rescan.data());
        hipLaunchKernelGGL(k_lv_big_items, g_v, kBlock, 0, h.stream, (uint32_t const*)L.off.data(), (uint32_t const*)pos.data(), nv, LVM_MAX, kLongRow, big_items.data(),
                           mid_count.data() + 2, big_rows.data(), big_rescan.data());
        h.read_back(&big_rescan_h, (unsigned long long const*)big_rescan.data(), 1);
      }
      h.read_back(n_mid, mid_count.data(), 5);
      if (n_mid[2]) { big_bits.resize_discard(n_mid[2]); big_c.resize_discard(n_mid[2]); }
    }
    uint32_t nh = 0;
    h.read_back(&nh, pos.data() + ne, 1);
    n_sorted = nh;
    size_t const h1 = (size_t)std::max<int64_t>(n_sorted, 1);
    Lh.src.resize_discard(h1); Lh.dst.resize_discard(h1); Lh.w.resize_discard(h1);
    hrow0.resize_discard(h1);
    if (n_sorted > 0)
      hipLaunchKernelGGL(k_lv_hub_compact, g_e, kBlock, 0, h.stream, (int32_t const*)L.src.data(), (int32_t const*)L.dst.data(), (double const*)L.w.data(),
                         (uint32_t const*)L.off.data(), (uint32_t const*)flag.data(), (uint32_t const*)pos.data(), ne, Lh.src.data(), Lh.dst.data(), Lh.w.data(),
                         hrow0.data());
    h.sync();
  }
  // hub rows: the sorted path (default), or a global hash table (CUGRAPH_AMD_LOUVAIN_HUB=hash; measured SLOWER -- 0.129 s against
  // 0.114 s at RMAT-22, 2.43 s against 2.10 s at RMAT-26: an insertion is two device-scope atomics per run of equal clusters and
  // k_lv_hub_insert manages 7 G edges/s, where the radix passes stream)
  bool const hub_hash = use_hash && n_sorted > 0 && getenv("CUGRAPH_AMD_LOUVAIN_HUB") && std::string(getenv("CUGRAPH_AMD_LOUVAIN_HUB")) == "hash";
  dvec<unsigned long long> hub_keys, hub_sums;
  if (hub_hash) { hub_keys.resize_discard((size_t)2 * n_sorted); hub_sums.resize_discard((size_t)2 * n_sorted); }
  int32_t const* const s_src = use_hash ? Lh.src.data() : L.src.data();
  int32_t const* const s_dst = use_hash ? Lh.dst.data() : L.dst.data();
  double const* const s_w    = use_hash ? Lh.w.data() : L.w.data();
  int const g_s = grid_for(n_sorted, kBlock, 8192);
  bool big_hash = use_big && !hub_hash && n_mid[2] > 0;  // (cleared for the rest of the level if a table ever fills up)
  if (big_hash) {
    big_ewf.resize_discard((size_t)n_sorted); big_ecl.resize_discard((size_t)n_sorted); big_rowsub.resize_discard((size_t)nv);
    HIP_TRY(hipMemsetAsync(big_rowsub.data(), 0, (size_t)nv * sizeof(unsigned long long), h.stream));
    hipLaunchKernelGGL(k_lv_big_prep, grid_for(n_sorted, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)Lh.src.data(), (int32_t const*)Lh.dst.data(),
                       (double const*)Lh.w.data(), n_sorted, scale, big_ewf.data(), big_rowsub.data());
  }
  // the big rows' pairs are partitioned by item every sweep when the items would otherwise read them more than kPartitionFrom times over
  // (CUGRAPH_AMD_LOUVAIN_PARTITION=0 / 1: never / whenever possible)
  constexpr double kPartitionFrom = 8.0;  // (RMAT-22's third level, 6.5: 12.6 ms partitioned against 11.1; RMAT-26's levels, 9.2 ... 34: 1.40 -> 1.30 s for the call)
  char const* env_part = getenv("CUGRAPH_AMD_LOUVAIN_PARTITION");
  bool const big_part = big_hash && n_mid[3] == 0 && n_mid[4] > 0 &&
                        (env_part ? atoi(env_part) != 0 : (double)big_rescan_h > kPartitionFrom * (double)std::max<int64_t>(n_sorted, 1));
  if (getenv("CUGRAPH_AMD_LOUVAIN_TRACE") && big_hash)
    fprintf(stderr, "[louvain]   big rows: %u rows, %u items, %lld edges, items would read them %.1f times over: %s\n", n_mid[4], n_mid[2], (long long)n_sorted,
            (double)big_rescan_h / (double)std::max<int64_t>(n_sorted, 1), big_part ? "partitioned by item every sweep" : "every item scans its row");
  if (big_part) {
    big_pcl.resize_discard((size_t)n_sorted); big_pwf.resize_discard((size_t)n_sorted); big_seg.resize_discard((size_t)n_mid[2]); big_rowself.resize_discard((size_t)nv);
  }
  // CUGRAPH_AMD_LOUVAIN_BIG_SLOTS=n (tests): an item gives up after n distinct clusters -- exercises the fall-back to the sorted path
  uint32_t const big_max_used = getenv("CUGRAPH_AMD_LOUVAIN_BIG_SLOTS") ? (uint32_t)std::max(1, atoi(getenv("CUGRAPH_AMD_LOUVAIN_BIG_SLOTS"))) : (uint32_t)(LVB_SLOTS - LVB_SLOTS / 8);
  size_t const n_keys = (size_t)std::max<int64_t>((hub_hash || big_hash) ? 1 : n_sorted, 1);  // buffers of the sorted path
  dvec<uint32_t> eperm(n_keys);
  dvec<uint64_t> ekeys(n_keys);
  segfix.resize_discard(n_keys);
  accepted.resize_discard((size_t)nv);
  HIP_TRY(hipMemsetAsync(kfix.data(), 0, (size_t)nv * sizeof(long long), h.stream));
  if (ne > 0) hipLaunchKernelGGL(k_vertex_weights, g_e, kBlock, 0, h.stream, (int32_t const*)L.src.data(), (double const*)L.w.data(), ne, scale,
                                 reinterpret_cast<unsigned long long*>(kfix.data()));
  if (mg) mg->merge_owned<long long>(kfix.data(), nv);  // a vertex's weight comes from its owner's edges
  hipLaunchKernelGGL(k_fix_to_double, g_v, kBlock, 0, h.stream, reinterpret_cast<unsigned long long const*>(kfix.data()), nv, 1.0 / scale, k.data());
  iota_i32(h, c.data(), nv, 0);
  iota_i32(h, accepted.data(), nv, 0);
  HIP_TRY(hipMemcpyAsync(afix.data(), kfix.data(), nv * sizeof(long long), hipMemcpyDeviceToDevice, h.stream));
  hipLaunchKernelGGL(k_fix_to_double, g_v, kBlock, 0, h.stream, (unsigned long long const*)afix.data(), nv, 1.0 / scale, a.data());
  int const vb = bits_of_u((uint64_t)std::max<int64_t>(nv - 1, 1));
  // One evaluation of the clustering c (update_clustering_by_delta_modularity, common_methods.cuh:259-447, up to the moves): the best move
  // of every vertex, the number of moves wanted in each direction, and -- as a by-product of the same tables: every row's weight into its own
  // cluster is in them -- the modularity of c itself (detail::compute_modularity, common_methods.cuh:176-228): the intra-cluster weight
  // as an exact fixed-point integer (order-free: the same bits on every path and with any number of ranks), the squared cluster weights
  // summed in 64 Ki chunks.  Round 4: the loop below therefore evaluates the clustering AFTER a sweep's moves before it decides whether
  // the sweep improved the modularity -- one evaluation per level is thrown away, and the separate pass over all edges after every sweep
  // (17 % of a call at RMAT-26: its gathers of c[dst] miss every cache) is gone.
  dvec<unsigned long long> stat(4);  // [0..1] three uint32: moves wanted down / up, "a big row's table filled up"; [2] intra-cluster weight; [3] sum of squares (double)
  uint32_t* const count           = reinterpret_cast<uint32_t*>(stat.data());
  unsigned long long* const ifix  = stat.data() + 2;
  int evals_in_level              = 0;
  // compute_louvain_min_vertex_move_gain with the reference's noise floor per weight type (common_methods.cuh:52-66)
  double const min_gain = std::max(threshold / (double)std::max<int64_t>(nv, 1), noise_floor);
  auto evaluate = [&]() {
    bool const sorted = n_sorted > 0 && !hub_hash && !big_hash;
    if (sorted) {
      hipLaunchKernelGGL(k_pair_keys, g_s, kBlock, 0, h.stream, s_src, (int32_t const*)nullptr, s_dst, (int32_t const*)c.data(), n_sorted, vb, ekeys.data(),
                         eperm.data());
      // first evaluation of a level: every vertex is its own cluster, so the key is (source, destination) -- the order the level's
      // edges are stored in (CSR order / the contraction's output): nothing to sort
      if (evals_in_level > 1) sort_pairs(h, ekeys, eperm, n_sorted, 2 * vb);
    }
    HIP_TRY(hipMemsetAsync(vfix.data(), 0, (size_t)nv * 3 * sizeof(unsigned long long), h.stream));  // selffix, subfix, best_bits
    HIP_TRY(hipMemsetAsync(best_c.data(), 0x7f, (size_t)nv * sizeof(int32_t), h.stream));
    HIP_TRY(hipMemsetAsync(stat.data(), 0, 3 * sizeof(unsigned long long), h.stream));  // counts, overflow flag, intra-cluster weight
    if (hub_hash) {
      HIP_TRY(hipMemsetAsync(hub_keys.data(), 0xFF, (size_t)2 * n_sorted * sizeof(unsigned long long), h.stream));
      HIP_TRY(hipMemsetAsync(hub_sums.data(), 0, (size_t)2 * n_sorted * sizeof(unsigned long long), h.stream));
      lv_hub_args HB{s_src, s_dst, s_w, hrow0.data(), L.off.data(), c.data(), k.data(), a.data(), m, resolution, scale, 1.0 / scale, n_sorted,
                     hub_keys.data(), hub_sums.data(), vfix.data(), vfix.data() + nv, vfix.data() + 2 * nv, best_c.data()};
      int const g_t = grid_for(2 * n_sorted, kBlock, 16384);
      hipLaunchKernelGGL(k_lv_hub_insert, g_s, kBlock, 0, h.stream, HB);
      hipLaunchKernelGGL(k_lv_hub_eval<0>, g_t, kBlock, 0, h.stream, HB);
      hipLaunchKernelGGL(k_lv_hub_eval<1>, g_t, kBlock, 0, h.stream, HB);
    }
    if (sorted) {
      HIP_TRY(hipMemsetAsync(segfix.data(), 0, (size_t)n_sorted * sizeof(unsigned long long), h.stream));
      lv_flat_args A{ekeys.data(), eperm.data(), s_dst, s_w, c.data(), k.data(), a.data(), m, resolution, scale, 1.0 / scale, n_sorted, vb,
                     segfix.data(), vfix.data(), vfix.data() + nv, vfix.data() + 2 * nv, best_c.data()};
      hipLaunchKernelGGL(k_segment_sums, g_s, kBlock, 0, h.stream, A);
      hipLaunchKernelGGL(k_segment_best<0>, g_s, kBlock, 0, h.stream, A);
      hipLaunchKernelGGL(k_segment_best<1>, g_s, kBlock, 0, h.stream, A);
    }
    if (use_hash && n_sorted < ne) {
      hipLaunchKernelGGL(k_lv_pack_ca, g_v, kBlock, 0, h.stream, (int32_t const*)c.data(), (double const*)a.data(), nv, ca.data());
      lv_hash_args HA{L.src.data(), L.dst.data(), L.w.data(), k.data(), ca.data(), m, resolution, scale, 1.0 / scale, ne,
                      vfix.data() + 2 * nv, best_c.data(), chunk_int.data(), chunk_range.data(), chunk_rs.data(), 0};
      int64_t const n_chunks = (ne + LVH_B - 1) / LVH_B;
      constexpr int64_t kChunksPerLaunch = ((int64_t)1 << 31) / LVH_THREADS;  // (grid x block must stay below 2^32 threads)
      for (HA.chunk0 = 0; HA.chunk0 < n_chunks; HA.chunk0 += kChunksPerLaunch)
        hipLaunchKernelGGL(k_lv_hash_chunks, (int)std::min<int64_t>(n_chunks - HA.chunk0, kChunksPerLaunch), LVH_THREADS, 0, h.stream, HA);
      HIP_TRY(hipGetLastError());  // (an oversized grid is refused at launch, silently otherwise)
      hipLaunchKernelGGL(k_lv_sum_u64, (int)std::max<int64_t>(1, std::min<int64_t>((n_chunks + 1023) / 1024, 256)), 1024, 0, h.stream, (unsigned long long const*)chunk_int.data(), n_chunks, ifix);
    }
    if (n_mid[0] + n_mid[1] > 0 || big_hash) {
      static bool attr_done = false;
      if (!attr_done) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<void const*>(k_lv_hash_rows<LVM_MAX * 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LVM_MAX * 2 * 12));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<void const*>(k_lv_hash_big<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LVB_SLOTS * 12));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<void const*>(k_lv_hash_big<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LVB_SLOTS * 12));
        attr_done = true;
      }
      lv_mid_args MA{nullptr, 0, L.dst.data(), L.off.data(), L.w.data(), c.data(), k.data(), a.data(), m, resolution, scale, 1.0 / scale,
                     vfix.data() + 2 * nv, best_c.data(), ifix};
      if (n_mid[0]) {
        MA.rows = mid_rows[0].data(); MA.n_rows = (int32_t)n_mid[0];
        hipLaunchKernelGGL(k_lv_hash_rows<LVM_MAX>, (int)std::min<uint32_t>(n_mid[0], (uint32_t)h.num_cus * 12), LVM_THREADS, LVM_MAX * 12, h.stream, MA);
      }
      if (n_mid[1]) {
        MA.rows = mid_rows[1].data(); MA.n_rows = (int32_t)n_mid[1];
        hipLaunchKernelGGL(k_lv_hash_rows<LVM_MAX * 2>, (int)std::min<uint32_t>(n_mid[1], (uint32_t)h.num_cus * 8), LVM_THREADS, LVM_MAX * 2 * 12, h.stream, MA);
      }
      if (big_hash) {
        lv_big_args BA{big_items.data(), (int32_t)n_mid[2], big_ecl.data(), big_ewf.data(), big_rowsub.data(), L.off.data(), c.data(), k.data(), a.data(), m, resolution,
                       scale, 1.0 / scale, vfix.data() + 2 * nv, best_c.data(), big_bits.data(), big_c.data(), count + 2, ifix, big_max_used, big_cursor.data(),
                       big_part ? big_seg.data() : (uint2*)nullptr, big_pcl.data(), big_pwf.data(), big_rowself.data()};
        HIP_TRY(hipMemsetAsync(big_cursor.data(), 0, 2 * sizeof(uint32_t), h.stream));
        hipLaunchKernelGGL(k_lv_big_gather, g_s, kBlock, 0, h.stream, s_dst, (int32_t const*)c.data(), n_sorted, big_ecl.data());
        if (big_part)
          hipLaunchKernelGGL(k_lv_big_partition, (int)std::min<uint32_t>(n_mid[4], (uint32_t)h.num_cus * 2), LVP_THREADS, 0, h.stream, (int4 const*)big_rows.data(), (int)n_mid[4],
                           (uint32_t const*)L.off.data(), (uint32_t const*)big_ecl.data(), (unsigned long long const*)big_ewf.data(), (int32_t const*)c.data(), big_pcl.data(),
                           big_pwf.data(), big_seg.data(), big_rowself.data(), big_cursor.data() + 1);
        if (big_part) hipLaunchKernelGGL(k_lv_hash_big<true>, (int)std::min<uint32_t>(n_mid[2], (uint32_t)h.num_cus * 8), LVB_THREADS, LVB_SLOTS * 12, h.stream, BA);
        else hipLaunchKernelGGL(k_lv_hash_big<false>, (int)std::min<uint32_t>(n_mid[2], (uint32_t)h.num_cus * 8), LVB_THREADS, LVB_SLOTS * 12, h.stream, BA);
        hipLaunchKernelGGL(k_lv_big_ties, grid_for((int64_t)n_mid[2], kBlock, 1024), kBlock, 0, h.stream, BA);
      }
    }
    if (sorted || hub_hash)  // these paths leave the rows' own-cluster weights in selffix[] (zero for the rows of the LDS paths)
      hipLaunchKernelGGL(k_lv_sum_u64, (int)std::max<int64_t>(1, std::min<int64_t>((nv + 1023) / 1024, 1024)), 1024, 0, h.stream, (unsigned long long const*)vfix.data(), nv, ifix);
    chunked_sum(h, nv, parts, reinterpret_cast<double*>(stat.data() + 3), [&](int np, double* p) { hipLaunchKernelGGL(k_part_squares, np, 1024, 0, h.stream, (double const*)a.data(), nv, p); });
    hipLaunchKernelGGL(k_best_finalize, g_v, kBlock, 0, h.stream, (unsigned long long const*)(vfix.data() + 2 * nv), best_c.data(), best_d.data(), nv);
    hipLaunchKernelGGL(k_count_moves, (int)std::max<int64_t>(1, std::min<int64_t>((nv + 1023) / 1024, 512)), 1024, 0, h.stream, (int32_t const*)c.data(),
                       (int32_t const*)best_c.data(), (double const*)best_d.data(), min_gain, nv, count);
  };
  struct eval_t { uint32_t moves[2]; double q; };
  auto run_eval = [&]() {
    ++evals_in_level;
    evaluate();
    unsigned long long got[4];
    h.read_back(got, stat.data(), 4);
    uint32_t cnt[4];
    std::memcpy(cnt, got, sizeof(cnt));
    if (cnt[2]) {  // a big row's LDS table filled up: the sorted path takes the big rows for the rest of this level
      big_hash = false;
      size_t const nk = (size_t)std::max<int64_t>(n_sorted, 1);
      eperm.resize_discard(nk); ekeys.resize_discard(nk); segfix.resize_discard(nk);
      evaluate();
      h.read_back(got, stat.data(), 4);
      std::memcpy(cnt, got, sizeof(cnt));
    }
    eval_t r{{cnt[0], cnt[1]}, 0.0};
    unsigned long long internal = got[2];
    if (mg) { mg->sum_u32x2(r.moves); internal = mg->sum_u64(internal); }  // (the cluster weights are complete on every rank)
    double squares;
    std::memcpy(&squares, &got[3], sizeof(squares));
    r.q = ((double)(long long)internal * (1.0 / scale)) / m - (resolution * squares) / (m * m);
    return r;
  };
  eval_t ev    = run_eval();
  double new_q = ev.q;
  double cur_q = new_q - 1.0;
  bool up_down = true;
  while (new_q > cur_q + threshold) {
    cur_q = new_q;
    ++st.sweeps;
    ++st.sweeps_in_level;
    // update_clustering_by_delta_modularity takes up_down BY VALUE (detail/common_methods.cuh:277, 445): a sweep without a move in its direction applies the
    // moves of the other direction, and that flip does not outlive the sweep -- the loop's own toggle below starts from the direction the sweep was given
    bool const dir = ev.moves[up_down ? 1 : 0] == 0 ? !up_down : up_down;
    // the moves + compute_cluster_keys_and_values: cluster weights of the new clustering
    hipLaunchKernelGGL(k_apply_moves, (int)std::max<int64_t>(1, std::min<int64_t>((nv + LVA_THREADS - 1) / LVA_THREADS, (int64_t)h.num_cus * 2)), LVA_THREADS, 0, h.stream, c.data(),
                       (int32_t const*)best_c.data(), (double const*)best_d.data(), min_gain, dir ? 1 : 0, nv, (long long const*)kfix.data(), afix.data());
    if (mg) {  // every rank moved ITS vertices: merge the labels, rebuild the cluster weights from them (integers: the same values the
               // single-GPU run reaches by adding and subtracting)
      mg->merge_owned<int32_t>(c.data(), nv);
      HIP_TRY(hipMemsetAsync(afix.data(), 0, (size_t)nv * sizeof(unsigned long long), h.stream));
      hipLaunchKernelGGL(k_cluster_weights, g_v, kBlock, 0, h.stream, (int32_t const*)c.data(), (long long const*)kfix.data(), nv, afix.data());
    }
    hipLaunchKernelGGL(k_fix_to_double, g_v, kBlock, 0, h.stream, (unsigned long long const*)afix.data(), nv, 1.0 / scale, a.data());
    up_down = !up_down;
    ev      = run_eval();
    new_q   = ev.q;
    if (new_q > cur_q + threshold) HIP_TRY(hipMemcpyAsync(accepted.data(), c.data(), nv * sizeof(int32_t), hipMemcpyDeviceToDevice, h.stream));
  }
  h.sync();
  return cur_q;
}


// ---- the partitioned driver: level 0 from this rank's slice, levels, contraction with one all-to-all per level
__global__ void k_lv_route(int32_t const* s, int32_t const* d, int64_t m, int64_t vmin, uint32_t const* vrank, int P, int32_t* owner, int32_t* vs, int32_t* vd)
{
  LV_LOOP(i, m)
  {
    int32_t const a = (int32_t)vrank[(int64_t)s[i] - vmin], b = (int32_t)vrank[(int64_t)d[i] - vmin];
    vs[i] = a; vd[i] = b; owner[i] = a % P;
  }
}
__global__ void k_lv_owner_of(int32_t const* src, int64_t n, int P, int32_t* owner) { LV_LOOP(i, n) owner[i] = src[i] % P; }
__global__ void k_lv_ext_of(uint32_t const* present, uint32_t const* vrank, int64_t vrange, int64_t vmin, int P, int rank, int32_t* ext_own)
{
  LV_LOOP(r, vrange) if (present[r]) { uint32_t const v = vrank[r]; if ((int)(v % (uint32_t)P) == rank) ext_own[v / (uint32_t)P] = (int32_t)(r + vmin); }
}
__global__ void k_lv_own_iota(int32_t* part, int64_t n, int P, int rank) { LV_LOOP(i, n) part[i] = (int32_t)(i * P + rank); }
template <typename T>
__global__ void k_lv_gather(T const* in, uint32_t const* perm, int64_t n, T* out) { LV_LOOP(i, n) out[i] = in[perm[i]]; }

// (src, dst, w) in any order -> sorted by (src, dst): the order the single-GPU level stores its edges in
void lv_sort_edges(handle_t const& h, int32_t const* src, int32_t const* dst, double const* w, int64_t n, int64_t nv, level_t& L)
{
  size_t const n1 = (size_t)std::max<int64_t>(n, 1);
  L.ne = n;
  L.src.resize_discard(n1); L.dst.resize_discard(n1); L.w.resize_discard(n1);
  if (n <= 0) return;
  int const vb = bits_of_u((uint64_t)std::max<int64_t>(nv - 1, 1));
  dvec<uint64_t> keys(n1);
  dvec<uint32_t> perm(n1);
  int const g = grid_for(n, kBlock, 8192);
  hipLaunchKernelGGL(k_pair_keys, g, kBlock, 0, h.stream, src, (int32_t const*)nullptr, dst, (int32_t const*)nullptr, n, vb, keys.data(), perm.data());
  sort_pairs(h, keys, perm, n, 2 * vb);
  hipLaunchKernelGGL(k_lv_gather<int32_t>, g, kBlock, 0, h.stream, src, (uint32_t const*)perm.data(), n, L.src.data());
  hipLaunchKernelGGL(k_lv_gather<int32_t>, g, kBlock, 0, h.stream, dst, (uint32_t const*)perm.data(), n, L.dst.data());
  hipLaunchKernelGGL(k_lv_gather<double>, g, kBlock, 0, h.stream, w, (uint32_t const*)perm.data(), n, L.w.data());
  h.sync();
}

}  // namespace

clustering_result_t* mg_run_louvain(handle_t const& h, graph_t& g, size_t max_level, double threshold, double resolution)
{
  HIP_TRY(hipSetDevice(h.device));
  mg_graph_t& mg = *g.mg;
  comm_t& c      = *mg.comm;
  CGA_EXPECTS(handle_comm(h) == mg.comm, CUGRAPH_INVALID_HANDLE, "multi-GPU Louvain: the handle is not on the communicator the graph was created on");
  {
    uint64_t scalars[3] = {(uint64_t)max_level, 0, 0};
    std::memcpy(&scalars[1], &threshold, sizeof(double));
    std::memcpy(&scalars[2], &resolution, sizeof(double));
    mg_agree_same(g, scalars, sizeof(scalars), "cugraph_louvain (max_level, threshold, resolution)");
  }
  int const P = c.size, me = c.rank;
  lv_mg_t M;
  M.c = &c; M.h = &h; M.P = P; M.rank = me; M.channel = 1;
  // level 0: vertex ids = rank of the external id among the graph's vertices (ascending): what a single-GPU graph created with
  // renumber = FALSE on dense ids runs on -- the partitioned clustering equals that run's vertex for vertex
  int64_t const nv0 = mg.nv_global;
  dvec<uint32_t> vrank((size_t)mg.vrange + 1);
  {
    dvec<uint32_t> pr((size_t)mg.vrange + 1);
    HIP_TRY(hipMemcpyAsync(pr.data(), mg.present.data(), (size_t)mg.vrange * 4, hipMemcpyDeviceToDevice, h.stream));
    HIP_TRY(hipMemsetAsync(pr.data() + mg.vrange, 0, 4, h.stream));
    exclusive_scan_u32(h, pr.data(), vrank.data(), mg.vrange + 1);
    h.sync();
  }
  int64_t const n_own = nv0 > me ? (nv0 - me + P - 1) / P : 0;
  level_t L;
  L.nv = nv0;
  {
    int64_t const m  = mg.el.n;
    size_t const m1  = (size_t)std::max<int64_t>(m, 1);
    dvec<int32_t> owner(m1), vs(m1), vd(m1);
    dvec<double> wd(m1);
    if (m > 0) {
      int const ge = grid_for(m, kBlock, 8192);
      hipLaunchKernelGGL(k_lv_route, ge, kBlock, 0, h.stream, (int32_t const*)mg.el.s.data(), (int32_t const*)mg.el.d.data(), m, mg.vmin, (uint32_t const*)vrank.data(), P, owner.data(),
                         vs.data(), vd.data());
      if (mg.el.wsize == 0) hipLaunchKernelGGL(k_to_double<float>, ge, kBlock, 0, h.stream, (float const*)nullptr, m, wd.data());  // constant weight 1 (louvain.cpp:86-92)
      else if (mg.el.wsize == 8) hipLaunchKernelGGL(k_to_double<double>, ge, kBlock, 0, h.stream, mg.el.w.as<double const>(), m, wd.data());
      else hipLaunchKernelGGL(k_to_double<float>, ge, kBlock, 0, h.stream, mg.el.w.as<float const>(), m, wd.data());
    }
    std::vector<dev_buf> got;
    int64_t const e_loc = mg_shuffle_by_owner(h, c, owner.data(), m, {{vs.data(), 4}, {vd.data(), 4}, {wd.data(), 8}}, got);
    CGA_EXPECTS(e_loc <= kMaxSignedEdges, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "multi-GPU Louvain: a rank's share must hold fewer than 2^31 edges");
    lv_sort_edges(h, got[0].as<int32_t const>(), got[1].as<int32_t const>(), got[2].as<double const>(), e_loc, nv0, L);
  }
  dvec<double> scal(1), parts;
  chunked_sum(h, L.ne, parts, scal.data(), [&](int np, double* p) { hipLaunchKernelGGL(k_part_sum, np, 1024, 0, h.stream, (double const*)L.w.data(), L.ne, p); });
  double m_loc = 0.0;
  h.read_back(&m_loc, (double const*)scal.data(), 1);
  double const m     = M.sum_f64(m_loc);  // compute_total_edge_weight over all ranks
  double const scale = fixed_scale(m);
  bool const trace   = getenv("CUGRAPH_AMD_LOUVAIN_TRACE") != nullptr;
  size_t const o1    = (size_t)std::max<int64_t>(n_own, 1);
  auto part          = std::make_unique<device_array_t>((size_t)n_own, INT32);
  dvec<int32_t> part_own(o1);
  hipLaunchKernelGGL(k_lv_own_iota, grid_for((int64_t)o1, kBlock, 8192), kBlock, 0, h.stream, part_own.data(), n_own, P, me);
  double best   = -1.0;
  size_t levels = 0;
  cugraph_amd_traversal_stats_t work{0, 0, 0, 0};
  while (levels < max_level && L.nv > 0 && m > 0.0) {
    ++levels;
    build_offsets(h, L);
    dvec<int32_t> cl;
    louvain_stats_t st;
    M.level_begin(L.nv);
    double const q = run_level(h, L, m, threshold, resolution, g.weight_type == FLOAT64 ? 1e-15 : 1e-12, cl, st, &M);
    M.level_end();
    if (trace && me == 0) fprintf(stderr, "[louvain mg] level %zu: %lld vertices, %lld local edges, %d sweeps, Q = %.9f\n", levels, (long long)L.nv, (long long)L.ne, st.sweeps, q);
    work.steps += (uint64_t)st.sweeps;
    work.edges_inspected += (uint64_t)st.sweeps * (uint64_t)L.ne;
    work.vertices_reached += (uint64_t)st.sweeps * (uint64_t)L.nv;
    work.edges_of_reached += (uint64_t)L.ne;
    if (q <= best) break;
    best = q;
    // graph_contraction: dense labels (every rank holds the full label vector: the same ranks everywhere), flattening of the owned vertices
    dvec<uint32_t> used((size_t)L.nv + 1), rank((size_t)L.nv + 1);
    HIP_TRY(hipMemsetAsync(used.data(), 0, ((size_t)L.nv + 1) * sizeof(uint32_t), h.stream));
    hipLaunchKernelGGL(k_mark_labels, grid_for(L.nv, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)cl.data(), L.nv, used.data());
    exclusive_scan_u32(h, used.data(), rank.data(), L.nv + 1);
    uint32_t ncl = 0;
    h.read_back(&ncl, rank.data() + L.nv, 1);
    hipLaunchKernelGGL(k_relabel, grid_for(L.nv, kBlock, 8192), kBlock, 0, h.stream, cl.data(), (uint32_t const*)rank.data(), L.nv);
    // coarse edges: local aggregation, all-to-all to the owner of the source cluster, second aggregation (integers: the single-GPU sums)
    int const cb = bits_of_u((uint64_t)std::max<int64_t>((int64_t)ncl - 1, 1));
    dvec<int32_t> lsrc, ldst;
    dvec<unsigned long long> lwfix;
    int64_t n_loc = 0;
    if (L.ne > 0) {
      dvec<uint64_t> keys((size_t)L.ne);
      dvec<uint32_t> perm((size_t)L.ne), head((size_t)L.ne + 1), pos((size_t)L.ne + 1);
      hipLaunchKernelGGL(k_pair_keys, grid_for(L.ne, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)L.src.data(), (int32_t const*)cl.data(), (int32_t const*)L.dst.data(),
                         (int32_t const*)cl.data(), L.ne, cb, keys.data(), perm.data());
      sort_pairs(h, keys, perm, L.ne, 2 * cb);
      hipLaunchKernelGGL(k_heads, grid_for(L.ne, kBlock, 8192), kBlock, 0, h.stream, (uint64_t const*)keys.data(), L.ne, head.data());
      HIP_TRY(hipMemsetAsync(head.data() + L.ne, 0, sizeof(uint32_t), h.stream));
      exclusive_scan_u32(h, head.data(), pos.data(), L.ne + 1);
      uint32_t nce = 0;
      h.read_back(&nce, pos.data() + L.ne, 1);
      n_loc = nce;
      size_t const n1 = (size_t)std::max<uint32_t>(nce, 1);
      lsrc.resize_discard(n1); ldst.resize_discard(n1); lwfix.resize_discard(n1);
      HIP_TRY(hipMemsetAsync(lwfix.data(), 0, n1 * sizeof(unsigned long long), h.stream));
      hipLaunchKernelGGL(k_coarse_edges, grid_for(L.ne, kBlock, 8192), kBlock, 0, h.stream, (uint64_t const*)keys.data(), (uint32_t const*)perm.data(), (uint32_t const*)head.data(),
                         (uint32_t const*)pos.data(), (double const*)L.w.data(), L.ne, cb, scale, lsrc.data(), ldst.data(), lwfix.data(), (unsigned long long const*)nullptr);
      h.sync();
    } else {
      lsrc.resize_discard(1); ldst.resize_discard(1); lwfix.resize_discard(1);
    }
    level_t N;
    N.nv = ncl;
    {
      dvec<int32_t> owner((size_t)std::max<int64_t>(n_loc, 1));
      if (n_loc > 0) hipLaunchKernelGGL(k_lv_owner_of, grid_for(n_loc, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)lsrc.data(), n_loc, P, owner.data());
      std::vector<dev_buf> got;
      int64_t const n_r = mg_shuffle_by_owner(h, c, owner.data(), n_loc, {{lsrc.data(), 4}, {ldst.data(), 4}, {lwfix.data(), 8}}, got);
      CGA_EXPECTS(n_r <= kMaxSignedEdges, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "multi-GPU Louvain: a rank's coarse share must hold fewer than 2^31 edges");
      if (n_r > 0) {
        dvec<uint64_t> keys((size_t)n_r);
        dvec<uint32_t> perm((size_t)n_r), head((size_t)n_r + 1), pos((size_t)n_r + 1);
        hipLaunchKernelGGL(k_pair_keys, grid_for(n_r, kBlock, 8192), kBlock, 0, h.stream, got[0].as<int32_t const>(), (int32_t const*)nullptr, got[1].as<int32_t const>(),
                           (int32_t const*)nullptr, n_r, cb, keys.data(), perm.data());
        sort_pairs(h, keys, perm, n_r, 2 * cb);
        hipLaunchKernelGGL(k_heads, grid_for(n_r, kBlock, 8192), kBlock, 0, h.stream, (uint64_t const*)keys.data(), n_r, head.data());
        HIP_TRY(hipMemsetAsync(head.data() + n_r, 0, sizeof(uint32_t), h.stream));
        exclusive_scan_u32(h, head.data(), pos.data(), n_r + 1);
        uint32_t nce = 0;
        h.read_back(&nce, pos.data() + n_r, 1);
        N.ne = nce;
        size_t const n1 = (size_t)std::max<uint32_t>(nce, 1);
        N.src.resize_discard(n1); N.dst.resize_discard(n1); N.w.resize_discard(n1);
        dvec<unsigned long long> cwfix(n1);
        HIP_TRY(hipMemsetAsync(cwfix.data(), 0, n1 * sizeof(unsigned long long), h.stream));
        hipLaunchKernelGGL(k_coarse_edges, grid_for(n_r, kBlock, 8192), kBlock, 0, h.stream, (uint64_t const*)keys.data(), (uint32_t const*)perm.data(), (uint32_t const*)head.data(),
                           (uint32_t const*)pos.data(), (double const*)nullptr, n_r, cb, scale, N.src.data(), N.dst.data(), cwfix.data(), got[2].as<unsigned long long const>());
        hipLaunchKernelGGL(k_fix_to_double, grid_for((int64_t)n1, kBlock, 8192), kBlock, 0, h.stream, (unsigned long long const*)cwfix.data(), (int64_t)n1, 1.0 / scale, N.w.data());
        h.sync();
      } else {
        N.ne = 0;
        N.src.resize_discard(1); N.dst.resize_discard(1); N.w.resize_discard(1);
      }
    }
    // The coarse vertices in the reference's numbering (coarse_degree_order): a coarse vertex's edges are all with its owner now, so the owners' row
    // lengths ARE the degrees; every rank receives all of them (one merge), sorts them the same way, renames its coarse edges, and the edges move to
    // the owners of the NEW ids (a second, smaller all-to-all: the numbering decides the next level's ties, so it cannot stay the label order).
    {
      dvec<uint32_t> deg((size_t)ncl + 1), new_id, new_off;
      HIP_TRY(hipMemsetAsync(deg.data(), 0, ((size_t)ncl + 1) * sizeof(uint32_t), h.stream));
      if (N.ne > 0) {
        dvec<uint32_t> fl((size_t)ncl + 1);
        HIP_TRY(hipMemsetAsync(fl.data(), 0, ((size_t)ncl + 1) * sizeof(uint32_t), h.stream));
        hipLaunchKernelGGL(k_row_bounds, grid_for(N.ne, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)N.src.data(), N.ne, fl.data(), deg.data());
        hipLaunchKernelGGL(k_row_lengths, grid_for((int64_t)ncl, kBlock, 8192), kBlock, 0, h.stream, (uint32_t const*)fl.data(), deg.data(), (int64_t)ncl);
        h.sync();  // (`fl`)
      }
      M.level_begin((int64_t)ncl);
      M.merge_owned<uint32_t>(deg.data(), (int64_t)ncl);  // deg[v] <- the row length rank v % P holds
      M.level_end();
      coarse_degree_order(h, deg.data(), (int64_t)ncl, new_id, new_off);
      hipLaunchKernelGGL(k_relabel, grid_for(L.nv, kBlock, 8192), kBlock, 0, h.stream, cl.data(), (uint32_t const*)new_id.data(), L.nv);
      if (n_own > 0) hipLaunchKernelGGL(k_compose, grid_for(n_own, kBlock, 8192), kBlock, 0, h.stream, part_own.data(), (int32_t const*)cl.data(), n_own);
      size_t const n1 = (size_t)std::max<int64_t>(N.ne, 1);
      dvec<int32_t> owner(n1);
      if (N.ne > 0) {
        hipLaunchKernelGGL(k_relabel, grid_for(N.ne, kBlock, 8192), kBlock, 0, h.stream, N.src.data(), (uint32_t const*)new_id.data(), N.ne);
        hipLaunchKernelGGL(k_relabel, grid_for(N.ne, kBlock, 8192), kBlock, 0, h.stream, N.dst.data(), (uint32_t const*)new_id.data(), N.ne);
        hipLaunchKernelGGL(k_lv_owner_of, grid_for(N.ne, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)N.src.data(), N.ne, P, owner.data());
      }
      std::vector<dev_buf> got;
      int64_t const n_r = mg_shuffle_by_owner(h, c, owner.data(), N.ne, {{N.src.data(), 4}, {N.dst.data(), 4}, {N.w.data(), 8}}, got);
      CGA_EXPECTS(n_r <= kMaxSignedEdges, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "multi-GPU Louvain: a rank's coarse share must hold fewer than 2^31 edges");
      level_t R;
      R.nv = ncl;
      lv_sort_edges(h, got[0].as<int32_t const>(), got[1].as<int32_t const>(), got[2].as<double const>(), n_r, (int64_t)ncl, R);
      h.sync();
      N = std::move(R);
    }
    L = std::move(N);
  }
  const_cast<handle_t&>(h).last_stats = work;
  auto res        = std::make_unique<clustering_result_t>();
  res->modularity = best;
  res->vertices   = new device_array_t((size_t)n_own, INT32);
  if (n_own > 0) {
    hipLaunchKernelGGL(k_lv_ext_of, grid_for(mg.vrange, kBlock, 8192), kBlock, 0, h.stream, (uint32_t const*)mg.present.data(), (uint32_t const*)vrank.data(), mg.vrange, mg.vmin, P, me,
                       res->vertices->buf.as<int32_t>());
    HIP_TRY(hipMemcpyAsync(part->buf.ptr, part_own.data(), (size_t)n_own * 4, hipMemcpyDeviceToDevice, h.stream));
  }
  res->clusters = part.release();
  h.sync();
  c.check("multi-GPU Louvain");
  return res.release();
}

namespace {
}  // namespace
}  // namespace cga

using namespace cga;

extern "C" cugraph_error_code_t cugraph_louvain(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph, size_t max_level, double threshold,
                                                double resolution, bool_t /*do_expensive_check*/, cugraph_hierarchical_clustering_result_t** result,
                                                cugraph_error_t** error)
{
  if (result) *result = nullptr;
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    graph_t& g        = GM(graph);
    CGA_EXPECTS(result != nullptr, CUGRAPH_INVALID_INPUT, "result is NULL");
    HIP_TRY(hipSetDevice(h.device));
    if (g.mg) {  // a graph from cugraph_graph_create_mg on a communicator handle: collective
      clustering_result_t* r = mg_run_louvain(h, g, max_level, threshold, resolution);
      outer_replace_ids(h, g, r->vertices);  // INT64 ids of a multi-GPU graph: back in the caller's id space (no-op otherwise)
      *result = reinterpret_cast<cugraph_hierarchical_clustering_result_t*>(r);
      return;
    }
    // (round 6: edge positions are unsigned words on the single-GPU path, as in graph construction and the traversals: up to kMaxGraphEdges, which creation enforces)
    ensure_orientation(h, g, false);  // louvain expects store_transposed == false (louvain.cpp:60-66)
    orientation_t const& o = g.csr;
    int64_t const nv0 = g.nv;
    level_t L;
    L.nv = nv0;
    L.ne = g.ne;
    size_t const e1 = (size_t)std::max<int64_t>(L.ne, 1);
    L.src.resize_discard(e1); L.dst.resize_discard(e1); L.w.resize_discard(e1);
    if (L.ne > 0) {
      hipLaunchKernelGGL(k_expand_src, grid_for(nv0 * 16, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)o.offsets.data(), nv0, L.src.data());
      HIP_TRY(hipMemcpyAsync(L.dst.data(), o.indices.data(), L.ne * sizeof(int32_t), hipMemcpyDeviceToDevice, h.stream));
      L.off.resize_discard((size_t)nv0 + 1);
      HIP_TRY(hipMemcpyAsync(L.off.data(), o.offsets.data(), ((size_t)nv0 + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, h.stream));
      L.have_off = true;
      int const ge = grid_for(L.ne, kBlock, 8192);
      if (!g.has_weights) hipLaunchKernelGGL(k_to_double<float>, ge, kBlock, 0, h.stream, (float const*)nullptr, L.ne, L.w.data());  // constant weight 1 (louvain.cpp:86-92)
      else if (g.weight_type == FLOAT64) hipLaunchKernelGGL(k_to_double<double>, ge, kBlock, 0, h.stream, o.weights.as<double const>(), L.ne, L.w.data());
      else hipLaunchKernelGGL(k_to_double<float>, ge, kBlock, 0, h.stream, o.weights.as<float const>(), L.ne, L.w.data());
    }
    dvec<double> scal(1), parts;
    chunked_sum(h, L.ne, parts, scal.data(), [&](int np, double* p) { hipLaunchKernelGGL(k_part_sum, np, 1024, 0, h.stream, (double const*)L.w.data(), L.ne, p); });
    double m = 0.0;  // compute_total_edge_weight
    h.read_back(&m, (double const*)scal.data(), 1);
    double const scale = fixed_scale(m);
    bool const trace = getenv("CUGRAPH_AMD_LOUVAIN_TRACE") != nullptr;
    auto const t_begin = std::chrono::steady_clock::now();
    auto part      = std::make_unique<device_array_t>((size_t)nv0, g.vertex_type);
    iota_i32(h, part->buf.as<int32_t>(), nv0, 0);
    double best = -1.0;
    size_t levels = 0;
    // work done (cugraph_amd_last_traversal_stats): steps = sweeps, edges_inspected = sum over sweeps of the level's edges,
    // vertices_reached = sum over sweeps of the level's vertices, edges_of_reached = sum over levels of the level's edges (contractions)
    cugraph_amd_traversal_stats_t work{0, 0, 0, 0};
    while (levels < max_level && L.nv > 0 && m > 0.0) {
      ++levels;
      build_offsets(h, L);
      dvec<int32_t> c;
      louvain_stats_t st;
      auto const t_level = std::chrono::steady_clock::now();
      double const q = run_level(h, L, m, threshold, resolution, g.weight_type == FLOAT64 ? 1e-15 : 1e-12, c, st);
      if (trace)
        fprintf(stderr, "[louvain] level %zu: %lld vertices, %lld edges, %d sweeps, Q = %.9f, %.1f ms\n", levels, (long long)L.nv, (long long)L.ne, st.sweeps, q,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_level).count());
      work.steps += (uint64_t)st.sweeps;
      work.edges_inspected += (uint64_t)st.sweeps * (uint64_t)L.ne;
      work.vertices_reached += (uint64_t)st.sweeps * (uint64_t)L.nv;
      work.edges_of_reached += (uint64_t)L.ne;
      if (q <= best) break;
      best = q;
      // graph_contraction (common_methods.cuh:230-257): dense labels, flattening, coarse edges
      dvec<uint32_t> used((size_t)L.nv + 1), rank((size_t)L.nv + 1);
      HIP_TRY(hipMemsetAsync(used.data(), 0, ((size_t)L.nv + 1) * sizeof(uint32_t), h.stream));
      hipLaunchKernelGGL(k_mark_labels, grid_for(L.nv, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)c.data(), L.nv, used.data());
      exclusive_scan_u32(h, used.data(), rank.data(), L.nv + 1);
      uint32_t ncl = 0;
      h.read_back(&ncl, rank.data() + L.nv, 1);
      hipLaunchKernelGGL(k_relabel, grid_for(L.nv, kBlock, 8192), kBlock, 0, h.stream, c.data(), (uint32_t const*)rank.data(), L.nv);
      level_t N;
      N.nv = ncl;
      dvec<uint32_t> new_id, new_off;
      if (L.ne > 0) {
        dvec<uint64_t> keys((size_t)L.ne);
        dvec<uint32_t> perm((size_t)L.ne), head((size_t)L.ne + 1), pos((size_t)L.ne + 1);
        int const cb = bits_of_u((uint64_t)std::max<int64_t>((int64_t)ncl - 1, 1));
        hipLaunchKernelGGL(k_pair_keys, grid_for(L.ne, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)L.src.data(), (int32_t const*)c.data(),
                           (int32_t const*)L.dst.data(), (int32_t const*)c.data(), L.ne, cb, keys.data(), perm.data());
        sort_pairs(h, keys, perm, L.ne, 2 * cb);
        hipLaunchKernelGGL(k_heads, grid_for(L.ne, kBlock, 8192), kBlock, 0, h.stream, (uint64_t const*)keys.data(), L.ne, head.data());
        HIP_TRY(hipMemsetAsync(head.data() + L.ne, 0, sizeof(uint32_t), h.stream));
        exclusive_scan_u32(h, head.data(), pos.data(), L.ne + 1);
        uint32_t nce = 0;
        h.read_back(&nce, pos.data() + L.ne, 1);
        N.ne = nce;
        size_t const n1 = (size_t)std::max<uint32_t>(nce, 1);
        N.src.resize_discard(n1); N.dst.resize_discard(n1); N.w.resize_discard(n1);
        dvec<unsigned long long> cwfix(n1);
        HIP_TRY(hipMemsetAsync(cwfix.data(), 0, n1 * sizeof(unsigned long long), h.stream));
        hipLaunchKernelGGL(k_coarse_edges, grid_for(L.ne, kBlock, 8192), kBlock, 0, h.stream, (uint64_t const*)keys.data(), (uint32_t const*)perm.data(),
                           (uint32_t const*)head.data(), (uint32_t const*)pos.data(), (double const*)L.w.data(), L.ne, cb, scale, N.src.data(), N.dst.data(),
                           cwfix.data());
        // the coarse vertices in the reference's numbering (coarse_degree_order): degrees = the rows' lengths, the rows move, the weights leave fixed point
        build_offsets(h, N);  // (rows of the label-order ids)
        dvec<uint32_t> deg((size_t)ncl + 1);
        hipLaunchKernelGGL(k_lv_row_lengths_of, grid_for((int64_t)ncl, kBlock, 8192), kBlock, 0, h.stream, (uint32_t const*)N.off.data(), (int64_t)ncl, deg.data());
        coarse_degree_order(h, deg.data(), (int64_t)ncl, new_id, new_off);
        dvec<int32_t> src2(n1), dst2(n1);
        hipLaunchKernelGGL(k_lv_renumber_rows, grid_for((int64_t)nce, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)N.src.data(), (int32_t const*)N.dst.data(),
                           (unsigned long long const*)cwfix.data(), (int64_t)nce, (uint32_t const*)N.off.data(), (uint32_t const*)new_off.data(), (uint32_t const*)new_id.data(),
                           1.0 / scale, src2.data(), dst2.data(), N.w.data());
        HIP_TRY(hipMemcpyAsync(N.off.data(), new_off.data(), ((size_t)ncl + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, h.stream));
        h.sync();
        N.src = std::move(src2);
        N.dst = std::move(dst2);
      } else {
        N.ne = 0;
        N.src.resize_discard(1); N.dst.resize_discard(1); N.w.resize_discard(1);
        dvec<uint32_t> deg((size_t)ncl + 1);  // no edges: every degree is zero, the numbering stays the label order
        HIP_TRY(hipMemsetAsync(deg.data(), 0, ((size_t)ncl + 1) * sizeof(uint32_t), h.stream));
        coarse_degree_order(h, deg.data(), (int64_t)ncl, new_id, new_off);
      }
      hipLaunchKernelGGL(k_relabel, grid_for(L.nv, kBlock, 8192), kBlock, 0, h.stream, c.data(), (uint32_t const*)new_id.data(), L.nv);
      hipLaunchKernelGGL(k_compose, grid_for(nv0, kBlock, 8192), kBlock, 0, h.stream, part->buf.as<int32_t>(), (int32_t const*)c.data(), nv0);
      h.sync();
      L = std::move(N);
    }
    if (trace)
      fprintf(stderr, "[louvain] %zu levels, Q = %.9f, %.1f ms\n", levels, best,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    const_cast<handle_t&>(h).last_stats = work;
    auto res        = std::make_unique<clustering_result_t>();
    res->modularity = best;
    res->vertices   = new device_array_t((size_t)nv0, g.vertex_type);
    if (nv0 > 0) HIP_TRY(hipMemcpyAsync(res->vertices->buf.ptr, g.number_map.data(), nv0 * sizeof(int32_t), hipMemcpyDeviceToDevice, h.stream));
    res->clusters = part.release();
    h.sync();
    outer_replace_ids(h, g, res->vertices);
    if (g.outer.active && g.outer.type == INT64) {  // cluster ids carry the vertex type (louvain.cpp:24-135): widen
      outer_ids_t widen;
      widen.active = true; widen.identity = true; widen.type = INT64;
      device_array_t* wide = outer_from_compact(h, widen, res->clusters->buf.as<int32_t>(), (int64_t)res->clusters->size);
      h.sync();
      delete res->clusters;
      res->clusters = wide;
    }
    *result = reinterpret_cast<cugraph_hierarchical_clustering_result_t*>(res.release());
  });
}

static cugraph_type_erased_device_array_view_t* lv_view(device_array_t* a)
{
  return a ? reinterpret_cast<cugraph_type_erased_device_array_view_t*>(a->new_view()) : nullptr;
}
extern "C" cugraph_type_erased_device_array_view_t* cugraph_hierarchical_clustering_result_get_vertices(cugraph_hierarchical_clustering_result_t* r)
{
  return lv_view(reinterpret_cast<clustering_result_t*>(r)->vertices);
}
extern "C" cugraph_type_erased_device_array_view_t* cugraph_hierarchical_clustering_result_get_clusters(cugraph_hierarchical_clustering_result_t* r)
{
  return lv_view(reinterpret_cast<clustering_result_t*>(r)->clusters);
}
extern "C" double cugraph_hierarchical_clustering_result_get_modularity(cugraph_hierarchical_clustering_result_t* r)
{
  return reinterpret_cast<clustering_result_t*>(r)->modularity;
}
extern "C" void cugraph_hierarchical_clustering_result_free(cugraph_hierarchical_clustering_result_t* r) { delete reinterpret_cast<clustering_result_t*>(r); }
