// BFS and SSSP for gfx950: a frontier engine built from one wave-cooperative edge-expansion kernel.
//
// Replaces:
//   cugraph_bfs / cugraph_sssp C API                      cpp/src/c_api/bfs.cpp:189, cpp/src/c_api/sssp.cpp:136
//   detail::bfs                                            cpp/src/traversal/bfs_impl.cuh:133-870
//   detail::sssp (near-far)                                cpp/src/traversal/sssp_impl.cuh:169-566
//   transform_reduce_if_v_frontier_outgoing_e_by_dst       cpp/include/cugraph/prims/transform_reduce_if_v_frontier_outgoing_e_by_dst.cuh:617-1127
//   extract_transform_if_v_frontier_e (3 kernels)          cpp/include/cugraph/prims/detail/extract_transform_if_v_frontier_e.cuh:127/309/422
//   update_v_frontier / vertex_frontier_t buckets          cpp/include/cugraph/prims/update_v_frontier.cuh:164-244, vertex_frontier.cuh:242-550
//
// The reference expands a frontier into an edge buffer, radix-sorts it by destination, reduces
// duplicates (`any` / `minimum`), then applies v_op and re-buckets -- several kernels, a sort of
// roughly 2-3x the edge bytes, and >= 3 host syncs per level.  Here the reduce-by-destination is done
// in place with device-scope atomics on the per-vertex state (32-bit visited words / distance bits),
// which is where the sort's output would be scattered anyway:
//   BFS   test prev-visited bit -> atomicOr new-visited bit (first setter enqueues, writes distance)
//         -> atomicMin(parent) : deterministic minimum-id parent (a valid instance of reduce_op::any,
//         bfs_impl.cuh:467; the reference's own test only validates parents, bfs_test.cpp:217-233).
//   SSSP  near-far (Davidson) with atomicMin on the order-preserving bit pattern of the non-negative
//         distance; strict relax new < min(d[v], cutoff) (sssp_impl.cuh:58-71).  The fixed point is unique,
//         so distances are bit-identical to Dijkstra.  Parents = lexicographic min (distance, parent)
//         (sssp_impl.cuh:334) recovered by one pass over the settled edges.
// Expansion kernel: a wavefront takes 64 frontier vertices; vertices with degree < 64 are flattened
// across the wave (wave64 prefix sum of degrees + per-edge owner search in LDS) so consecutive lanes read
// consecutive adjacency words; degree >= 64 rows are walked by the whole wave; degree >= 2048 rows are
// deferred to a second kernel in which the whole grid strides the adjacency list.
#include "common.hpp"
#include "mg_graph.hpp"

#include <chrono>
#include "traversal_common.hpp"
#include "traversal_bottom_up.hpp"

#include <cfloat>
#include <cmath>

namespace cga {

namespace {

// --------------------------------------------------------------------------------------------- BFS
// Level-synchronous, direction-optimising (Beamer): small frontiers are expanded top-down from a queue (push over the
// out-edges, visited bits claimed with atomicOr after a plain pre-test), large ones bottom-up (every unvisited vertex
// scans its in-neighbours -- ascending ids = hubs first under the degree-sorted numbering -- and stops at the first one
// in the frontier bitmap; one wavefront owns 64 consecutive vertices = two bitmap words, so there are no atomics at all).
// Replaces bfs_impl.cuh:133-870 incl. the bottom-up branch (per_v_transform_reduce_if_outgoing_e, :587-805 there).
// Distances do not depend on the direction.  The parent reported for v is, among its valid parents (in-neighbours one
// level up), the one with the smallest INTERNAL id -- i.e. the highest-degree one: top-down levels claim it with atomicMin
// (after a plain pre-test), bottom-up levels get it for free because neighbour lists are sorted by internal id and the scan
// stops at the first frontier member.  A valid instance of the reference's reduce_op::any (bfs_impl.cuh:467; its test only
// validates parents, bfs_test.cpp:217-233), deterministic, and identical whichever direction each level ran in.
struct bfs_state {
  int32_t* dist;
  int32_t* pred;               // INTERNAL id of the parent, INT32_MAX = none yet (mapped to external ids / -1 at the end); nullptr when not requested
  uint32_t const* vis_prev;    // visited as of the start of the level
  uint32_t* vis_new;           // cumulative
  int32_t* q_next;
  counters_t* cnt;
  int32_t const* out_offsets;  // degree sums for the direction heuristic
  int32_t const* in_offsets;
  int32_t next_depth;
};

struct bfs_visit {
  bfs_state s;
  wave_queue wq;
  unsigned long long acc_out{0}, acc_in{0};
  __device__ __forceinline__ void operator()(int32_t u, int32_t v, eoff_t)
  {
    uint32_t bit = 1u << (v & 31);
    bool fresh   = false;
    if (!(s.vis_prev[v >> 5] & bit)) {
      // pre-test of the cumulative word: hub destinations are claimed once and then skipped without an atomic
      // (a stale read only costs a redundant atomicOr)
      // (agent-scope load: L2-served AND retained.  A non-temporal load is L2-served too but its lines are evicted first -- the 2 MB
      // bitmap then keeps missing: a push-only BFS at RMAT-24 takes 5.5 ms with it, 4.55 ms with this: profiles/r3_sssp_rounds.txt)
      bool claimed = (__hip_atomic_load(&s.vis_new[v >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) != 0;
      if (!claimed) {
        uint32_t old = atomicOr(&s.vis_new[v >> 5], bit);
        fresh        = !(old & bit);
      }
      if (fresh) {
        s.dist[v] = s.next_depth;
        acc_out += (unsigned long long)(eoff(s.out_offsets, v + 1) - eoff(s.out_offsets, v));
        acc_in += (unsigned long long)(eoff(s.in_offsets, v + 1) - eoff(s.in_offsets, v));
      }
      if (s.pred && u < __hip_atomic_load(&s.pred[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&s.pred[v], u);  // minimum internal id among the frontier parents
    }
    wq.push(fresh, v);
  }
  __device__ __forceinline__ void flush()
  {
    wq.flush();
    unsigned long long a = acc_out, b = acc_in;
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if ((threadIdx.x & 63) == 0 && (a | b)) { counter_sums_t* r = cnt_replica(s.cnt); atomicAdd(&r->out_edges, a); atomicAdd(&r->in_edges, b); }
  }
};

struct keep_all { __device__ __forceinline__ bool operator()(int32_t) const { return true; } };

__global__ void __launch_bounds__(TV_BLOCK) k_bfs_expand(int32_t const* q, int64_t n, int32_t const* offsets, int32_t const* indices,
                                                         int32_t* bigq, bfs_state s, int32_t big_deg, int32_t seg)
{
  __shared__ wave_queue_storage<1> wqs;
  wqs.init();
  bfs_visit f{s, wave_queue(wqs, 0, s.q_next, &s.cnt->n_next)};
  expand_frontier(q, n, offsets, indices, bigq, s.cnt, keep_all{}, f, big_deg, nullptr, seg);  // (EX_U edges in flight per lane, as SSSP does: no gain here)
  f.flush();
}
__global__ void __launch_bounds__(TV_BLOCK) k_bfs_expand_big(int32_t const* bigq, int32_t const* offsets, int32_t const* indices, bfs_state s, int32_t seg)
{
  __shared__ wave_queue_storage<1> wqs;
  wqs.init();
  bfs_visit f{s, wave_queue(wqs, 0, s.q_next, &s.cnt->n_next)};
  expand_big(bigq, offsets, indices, s.cnt, f, nullptr, seg);
  f.flush();
}

// Parents of the vertices a TOP-DOWN level has just discovered, pulled from their in-edges (round 5): the first in-neighbour -- ascending
// internal id -- that was visited before the level started.  Such a neighbour sits exactly one level up (a shallower one would have given
// the vertex a smaller depth), so this is the rule of the push with atomicMin (smallest internal id among the frontier parents) and of the
// bottom-up levels, evaluated per DISCOVERED vertex (10^4-10^5 in the levels that run top-down) instead of per inspected edge: the push
// then does nothing for the parents (it reads pred[v] for every edge into a not yet visited vertex and atomicMins most of them:
// +0.2 ms per BFS at RMAT-24).  Measured slower than that (see run_bfs) and therefore opt-in; kept under test.  One wavefront per vertex, 64 neighbours per step, ballot early exit; rows above BFS_PULL_LONG in-edges go to
// a list that k_bfs_pull_parents_long scans with a workgroup per row (no single wavefront walks a 10^6-entry row).
constexpr int32_t BFS_PULL_LONG = 8192;
__global__ void __launch_bounds__(TV_BLOCK) k_bfs_pull_parents(int32_t const* q, counters_t* cnt, int32_t const* in_off, int32_t const* in_idx, uint32_t const* vis_prev,
                                                               int32_t* pred, int32_t* longq)
{
  uint32_t const n = cnt->n_next;  // the level's expansion kernels have completed (stream order): the queue's final size
  int const lane = threadIdx.x & 63;
  uint32_t const gwave = (uint32_t)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6), nwaves = (uint32_t)(((int64_t)gridDim.x * blockDim.x) >> 6);
  for (uint32_t i = gwave; i < n; i += nwaves) {
    int32_t const v = q[i];
    eoff_t const b = eoff(in_off, v), e = eoff(in_off, v + 1);
    if (e - b > (eoff_t)BFS_PULL_LONG) {
      if (lane == 0) longq[atomicAdd(&cnt->n_set, 1u)] = v;
      continue;
    }
    int32_t best = INT32_MAX;
    for (eoff_t p0 = b; p0 < e; p0 += 64) {
      eoff_t const p = p0 + lane;
      int32_t const u = p < e ? in_idx[p] : -1;
      bool const hit  = u >= 0 && ((vis_prev[(uint32_t)u >> 5] >> ((uint32_t)u & 31u)) & 1u);
      uint64_t const m = __ballot(hit);
      if (m) { best = __shfl(u, __ffsll((unsigned long long)m) - 1); break; }
    }
    if (lane == 0) pred[v] = best;
  }
}
__global__ void __launch_bounds__(TV_BLOCK) k_bfs_pull_parents_long(int32_t const* longq, counters_t const* cnt, int32_t const* in_off, int32_t const* in_idx,
                                                                    uint32_t const* vis_prev, int32_t* pred)
{  // one workgroup per long row at a time: its wavefronts take the row's 64-entry chunks round-robin (ascending), the position of the
   // earliest hit is kept in LDS and ends the scan of every wavefront that has passed it
  __shared__ uint32_t s_best;
  uint32_t const n = cnt->n_set;
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
    int32_t const v = longq[k];
    eoff_t const b = eoff(in_off, v), e = eoff(in_off, v + 1);
    if (threadIdx.x == 0) s_best = 0xFFFFFFFFu;
    __syncthreads();
    for (eoff_t p0 = b + (eoff_t)wave * 64; p0 < e; p0 += (eoff_t)TV_WAVES * 64) {
      if (p0 - b > *reinterpret_cast<volatile uint32_t*>(&s_best)) break;  // (wave-uniform: an LDS word)
      eoff_t const p = p0 + lane;
      int32_t const u = p < e ? in_idx[p] : -1;
      bool const hit  = u >= 0 && ((vis_prev[(uint32_t)u >> 5] >> ((uint32_t)u & 31u)) & 1u);
      uint64_t const m = __ballot(hit);
      if (m) {
        if (lane == 0) atomicMin(&s_best, (uint32_t)(p0 - b) + (uint32_t)(__ffsll((unsigned long long)m) - 1));
        break;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) pred[v] = s_best != 0xFFFFFFFFu ? in_idx[b + s_best] : INT32_MAX;
    __syncthreads();
  }
}

// front <- snapshot of the visited set (an unvisited vertex cannot have an in-neighbour that was visited before the
// latest level, so testing against everything visited so far is the same as testing against the frontier);
// vis_prev <- vis_new
__global__ void k_bfs_front_from_vis(uint32_t* vis_prev, uint32_t const* vis_new, uint32_t* front, int64_t nwords)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nwords; i += stride) {
    uint32_t n  = vis_new[i];
    front[i]    = n;
    vis_prev[i] = n;
  }
}

// queue <- set bits of a bitmap (order within the queue is immaterial)

// parents: internal ids -> external ids, INT32_MAX (none) -> -1 (invalid_vertex_id)
__global__ void k_bfs_finish_pred(int32_t* pred, int64_t n, int32_t const* labels)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    int32_t p = pred[i];
    pred[i]   = p == INT32_MAX ? -1 : (labels ? labels[p] : p);
  }
}

// result columns in one pass: ids <- the numbering; parents (when kept): internal ids -> external ids, INT32_MAX (none) -> -1
__global__ void __launch_bounds__(256) k_bfs_result_columns(int32_t const* number_map, int32_t* ids, int32_t* pred, int64_t n, int32_t const* labels)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    ids[i] = number_map[i];
    if (pred) {
      int32_t const p = pred[i];
      pred[i]         = p == INT32_MAX ? -1 : (labels ? labels[p] : p);
    }
  }
}

// dist / pred <- "unreached", the bitmaps and the counters <- 0
__global__ void __launch_bounds__(256) k_bfs_init_state(int32_t* dist, int32_t* pred, int64_t nv, uint32_t* vis_prev, uint32_t* vis_new, uint32_t* front, uint32_t* next,
                                                        int64_t nwords, counters_t* cnt)
{
  int64_t const t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  int64_t const n4 = nv / 4;
  int4 const big{INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX};
  for (int64_t i = t; i < n4; i += stride) {
    reinterpret_cast<int4*>(dist)[i] = big;
    if (pred) reinterpret_cast<int4*>(pred)[i] = big;
  }
  for (int64_t i = n4 * 4 + t; i < nv; i += stride) { dist[i] = INT32_MAX; if (pred) pred[i] = INT32_MAX; }
  for (int64_t i = t; i < nwords; i += stride) {
    vis_prev[i] = 0u; vis_new[i] = 0u;
    if (front) { front[i] = 0u; next[i] = 0u; }
  }
  uint32_t* c = reinterpret_cast<uint32_t*>(cnt);
  for (int64_t i = t; i < (int64_t)(sizeof(counters_t) / 4); i += stride) c[i] = 0u;
}

__global__ void k_bfs_init_sources(int32_t const* src, int64_t n, int32_t* dist, uint32_t* vis_prev, uint32_t* vis_new, int32_t* q,
                                   counters_t* cnt, int32_t const* out_offsets, int32_t const* in_offsets)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t v    = src[i];
  if (v < 0) { atomicAdd(&cnt->n_big, 1u); return; }  // not a vertex of the graph (renumber_ext_to_int): reported by the host after the read-back
  uint32_t bit = 1u << (v & 31);
  uint32_t old = atomicOr(&vis_new[v >> 5], bit);
  if (!(old & bit)) {  // duplicates in the source list are enqueued once
    atomicOr(&vis_prev[v >> 5], bit);
    dist[v] = 0;
    q[atomicAdd(&cnt->n_next, 1u)] = v;
    atomicAdd(&cnt->out_edges, (unsigned long long)(eoff(out_offsets, v + 1) - eoff(out_offsets, v)));
    atomicAdd(&cnt->in_edges, (unsigned long long)(eoff(in_offsets, v + 1) - eoff(in_offsets, v)));
  }
}

// ---------------------------------------------------------------------------------------------- SSSP
template <typename WT> struct dist_bits;
template <> struct dist_bits<float> {
  using type = uint32_t;
  static __device__ __forceinline__ uint32_t to(float x) { return __float_as_uint(x); }
  static __device__ __forceinline__ float from(uint32_t b) { return __uint_as_float(b); }
};
template <> struct dist_bits<double> {
  using type = unsigned long long;
  static __device__ __forceinline__ unsigned long long to(double x) { return (unsigned long long)__double_as_longlong(x); }
  static __device__ __forceinline__ double from(unsigned long long b) { return __longlong_as_double((long long)b); }
};

template <typename WT>
struct sssp_state {
  using bits_t = typename dist_bits<WT>::type;
  bits_t* dist;        // bit pattern of the (non-negative) tentative distance
  WT const* weights;
  int32_t* q_next;     // next near frontier
  int32_t* far;        // far pile
  uint32_t* mark_near; // last relax round in which the vertex entered q_next
  uint32_t* mark_far;  // last far epoch in which the vertex entered the far pile
  int32_t* q_set;      // light / heavy buckets: every vertex that entered the near frontier of the current bucket, once (nullptr: off)
  uint32_t* mark_set;  // ... its membership marks (set epoch)
  uint32_t set_epoch;
  counters_t* cnt;
  WT threshold;        // near / far split
  WT cutoff;
  uint32_t round;
  uint32_t far_epoch;
  int32_t const* out_offsets;  // CSR offsets: the out-degrees of the vertices that enter the near frontier are summed (counters_t::out_edges):
                               // the host knows the next round's edge count without a pass over the queue (pull rounds are chosen by it)
  // fp32 with predecessors (sssp_relax<WT, true>): (distance bits << 32 | external id of the parent) per vertex, lowered by ONE 64-bit
  // atomicMin -- the reference's reduction, a lexicographic minimum over (distance, predecessor) (sssp_impl.cuh:334), in the relaxation itself
  // instead of a sweep over the settled edges afterwards; `dist` is not used then
  unsigned long long* pk{nullptr};
  int32_t const* labels{nullptr};  // internal -> external id (nullptr: identity)
  int32_t source{-1};              // keeps its parent -1 whatever reaches it at distance 0
};
__device__ __forceinline__ uint32_t pk_dist_bits(unsigned long long const* pk, int32_t v) { return reinterpret_cast<uint32_t const*>(pk)[2 * (size_t)v + 1]; }

template <typename WT, bool PK = false>
struct sssp_relax {
  static_assert(!PK || sizeof(WT) == 4, "the packed (distance, parent) word holds an fp32 distance");
  sssp_state<WT> s;
  wave_queue wq_near, wq_far, wq_set;
  unsigned long long deg_acc{0};
  __device__ __forceinline__ void count_near(bool near, int32_t v)
  {
    if (near && s.out_offsets) deg_acc += (unsigned long long)(eoff(s.out_offsets, v + 1) - eoff(s.out_offsets, v));
  }
  __device__ __forceinline__ void flush()
  {
    wq_near.flush(); wq_far.flush(); wq_set.flush();
    unsigned long long a = deg_acc;
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if ((threadIdx.x & 63) == 0 && a) atomicAdd(&cnt_replica(s.cnt)->out_edges, a);
  }
  __device__ __forceinline__ void operator()(int32_t u, int32_t v, eoff_t p)
  {
    if constexpr (PK) {  // (the one-edge-at-a-time expansion: the phased form below, called in sequence)
      cand_t const c = pre(u, v, p);
      tok_t const t  = mid(v, c);
      post(u, v, c, t, mid2(v, c, t));
      return;
    }
    using B  = dist_bits<WT>;
    WT du    = B::from(s.dist[u]);
    WT nd    = du + s.weights[p];
    bool near = false, far = false, fresh = false;
    // agent-scope load: bypasses the per-CU vector cache, so once a hub has been lowered the other relaxations of this
    // round see it and skip the atomic (a plain load keeps reading the stale value from L1 and every edge into the hub
    // issues an atomicMin)
    if (nd < s.cutoff && nd < B::from(__hip_atomic_load(&s.dist[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
      auto old = atomicMin(&s.dist[v], B::to(nd));
      if (B::to(nd) < old) {  // this relaxation lowered d[v]
        if (nd < s.threshold) {
          near = atomicExch(&s.mark_near[v], s.round) != s.round;
          if (s.q_set && near) fresh = atomicExch(&s.mark_set[v], s.set_epoch) != s.set_epoch;
        } else {
          far = atomicExch(&s.mark_far[v], s.far_epoch) != s.far_epoch;
        }
      }
    }
    count_near(near, v);
    wq_near.push(near, v);
    wq_far.push(far, v);
    if (s.q_set) wq_set.push(fresh, v);  // (wave-uniform condition)
  }
  // the phased form of the same relaxation (expand_*_mlp: EX_U edges in flight per lane)
  struct cand_t { WT nd; bool pass; unsigned long long word; };  // word: the packed candidate (PK only)
  using tok_t = typename std::conditional<PK, unsigned long long, typename dist_bits<WT>::type>::type;
  __device__ __forceinline__ cand_t pre(int32_t u, int32_t v, eoff_t p) const
  {  // branch-free: the loads of the EX_U edges of a step go out together.  d[v] is read with an agent-scope load: L2-served, so once a
     // hub has been lowered the other relaxations of this round see it and skip the atomic (a non-temporal load is L2-served too, but
     // its lines are not retained: 14.0 ms instead of 10.4 per SSSP at RMAT-24)
    using B = dist_bits<WT>;
    if constexpr (PK) {
      int32_t const uu = u < 0 ? 0 : u, vv = v < 0 ? 0 : v;
      WT const nd = B::from(pk_dist_bits(s.pk, uu)) + s.weights[p];
      unsigned long long const word = ((unsigned long long)B::to(nd) << 32) | (uint32_t)(s.labels ? s.labels[uu] : uu);
      unsigned long long const cur  = __hip_atomic_load(&s.pk[vv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // strictly smaller (distance, parent): a shorter distance, or the same distance through a parent with a smaller external id
      bool const pass = (v >= 0) & (v != s.source) & (nd < s.cutoff) & (word < cur);
      return cand_t{nd, pass, word};
    } else {
      WT const nd = B::from(s.dist[u < 0 ? 0 : u]) + s.weights[p];
      WT const dv = B::from(__hip_atomic_load(&s.dist[v < 0 ? 0 : v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      bool const pass = (v >= 0) & (nd < s.cutoff) & (nd < dv);
      return cand_t{nd, pass, 0ull};
    }
  }
  __device__ __forceinline__ tok_t mid(int32_t v, cand_t c) const
  {
    using B = dist_bits<WT>;
    if constexpr (PK) return c.pass ? atomicMin(&s.pk[v], c.word) : ~0ull;
    else return c.pass ? atomicMin(&s.dist[v], B::to(c.nd)) : (tok_t)0;
  }
  using tok2_t = uint32_t;
  __device__ __forceinline__ tok2_t mid2(int32_t v, cand_t c, tok_t old) const
  {  // the relaxation lowered d[v]: claim the vertex for this round's near queue / this epoch's far pile
    using B = dist_bits<WT>;
    if constexpr (PK) {
      if (!(c.pass && (uint32_t)B::to(c.nd) < (uint32_t)(old >> 32))) return 0u;  // (a better parent at the same distance moves no queue)
    } else {
      if (!(c.pass && B::to(c.nd) < old)) return 0u;
    }
    bool const won = c.nd < s.threshold ? atomicExch(&s.mark_near[v], s.round) != s.round : atomicExch(&s.mark_far[v], s.far_epoch) != s.far_epoch;
    return won ? (c.nd < s.threshold ? 1u : 2u) : 0u;
  }
  __device__ __forceinline__ void post(int32_t, int32_t v, cand_t, tok_t, tok2_t won)
  {
    bool const near = won == 1u, far = won == 2u;
    bool fresh = false;
    if (s.q_set && near) fresh = atomicExch(&s.mark_set[v], s.set_epoch) != s.set_epoch;
    count_near(near, v);
    wq_near.push(near, v);
    wq_far.push(far, v);
    if (s.q_set) wq_set.push(fresh, v);
  }
};

// A PULL relaxation round: every row of the CSC scans its in-edges and relaxes the ones whose source sits in the frontier bitmap
// (sssp_impl.cuh:412-561 always pushes).  For the round right after the source -- a few ten thousand hubs whose out-edges are a third of the
// graph -- a push round is 79 M relaxations at 24-31 G/s (RMAT-24; most of them SUCCEED: an atomicMin, a mark exchange and a queue append
// each, on random lines); the pull round streams the in-edges once (8 bytes each), tests a bitmap that sits in L2, reads d[u] of the few
// frontier members from L2 and updates d[row] next to where its neighbours on the other lanes update it.  Same fixed point: relaxation
// order does not matter.  The adapter swaps the roles for sssp_relax: relax(u = in-neighbour, v = row, position in the CSC).
template <typename WT>
struct sssp_pull_fn {
  sssp_relax<WT> inner;
  uint32_t const* fbits;
  __device__ __forceinline__ void operator()(int32_t row, int32_t nbr, eoff_t p)
  {
    if ((fbits[(uint32_t)nbr >> 5] >> ((uint32_t)nbr & 31u)) & 1u) inner(nbr, row, p);
  }
  __device__ __forceinline__ void flush() { inner.flush(); }
};

__global__ void k_queue_to_bits(int32_t const* q, int64_t n, uint32_t* bits)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) { int32_t const v = q[i]; atomicOr(&bits[(uint32_t)v >> 5], 1u << ((uint32_t)v & 31u)); }
}

// The rows of fewer than big_deg in-edges: a wavefront takes 64 consecutive vertices; a lane scans its own row when it is shorter than 64
// (16-byte index loads, four probes of the frontier bitmap per step), longer rows are walked by the whole wavefront; a row's candidate
// distances are reduced in registers and the vertex is updated ONCE, by its only writer in this kernel -- no atomic on d[]: the edges stream
// at the rate of the bottom-up BFS levels instead of the 40-70 G edges/s of the frontier expansion.  Rows of big_deg or more in-edges go to
// bigq in BIG_SEG-edge segments for k_sssp_pull_big (which relaxes through sssp_relax: several workgroups share such a row).
template <typename WT>
__global__ void __launch_bounds__(TV_BLOCK) k_sssp_pull_rows(int32_t const* in_offsets, int32_t const* in_indices, WT const* in_weights, int64_t nv, uint32_t const* fbits,
                                                             int32_t* bigq, sssp_state<WT> s, int32_t big_deg)
{
  __shared__ wave_queue_storage<2> wqs;
  wqs.init();
  wave_queue wq_near(wqs, 0, s.q_next, &s.cnt->n_next), wq_far(wqs, 1, s.far, &s.cnt->n_far);
  using B              = dist_bits<WT>;
  int const lane       = threadIdx.x & 63;
  int64_t const gwave  = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int64_t const nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int64_t const ngroup = (nv + 63) >> 6;
  WT const inf         = std::numeric_limits<WT>::max();
  unsigned long long deg_acc = 0;
  auto in_front = [&](int32_t u) { return ((fbits[(uint32_t)u >> 5] >> ((uint32_t)u & 31u)) & 1u) != 0; };
  for (int64_t grp = gwave; grp < ngroup; grp += nwaves) {
    int64_t const v = grp * 64 + lane;
    eoff_t b    = 0;
    int32_t len = 0;
    if (v < nv) { b = eoff(in_offsets, v); len = (int32_t)(eoff(in_offsets, v + 1) - b); }
    bool const big = len >= big_deg;
    if (big) {
      uint32_t const nseg = ((uint32_t)len + BIG_SEG - 1) / BIG_SEG;
      uint32_t const at   = atomicAdd(&s.cnt->n_big, nseg);
      for (uint32_t sgm = 0; sgm < nseg; ++sgm) { bigq[2 * (at + sgm)] = (int32_t)v; bigq[2 * (at + sgm) + 1] = (int32_t)sgm; }
    }
    WT best = inf;
    bool const own = len > 0 && len < 64;
    int32_t longest = own ? len : 0;
    for (int o = 32; o > 0; o >>= 1) longest = max(longest, __shfl_xor(longest, o));
    for (int32_t k = 0; k < longest; k += BU_CHUNK) {
      int32_t u[BU_CHUNK];
      bu_load_chunk(in_indices, b + (eoff_t)k, own ? len - k : 0, u);
#pragma unroll
      for (int j = 0; j < BU_CHUNK; ++j)
        if (u[j] >= 0 && in_front(u[j])) best = min(best, B::from(s.dist[u[j]]) + in_weights[b + (eoff_t)(k + j)]);
    }
    uint64_t mid = __ballot(len >= 64 && !big);
    while (mid) {
      int const src = __ffsll((unsigned long long)mid) - 1;
      mid &= mid - 1;
      eoff_t const rb  = (eoff_t)__shfl((int)b, src);
      int32_t const rl = __shfl(len, src);
      WT m = inf;
      for (int32_t p = lane; p < rl; p += 64) {
        int32_t const u = in_indices[rb + (eoff_t)p];
        if (in_front(u)) m = min(m, B::from(s.dist[u]) + in_weights[rb + (eoff_t)p]);
      }
      for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
      if (lane == src) best = m;
    }
    bool near = false, far = false;
    if (v < nv && best < s.cutoff && best < B::from(s.dist[v])) {
      s.dist[v] = B::to(best);
      if (best < s.threshold) { near = atomicExch(&s.mark_near[v], s.round) != s.round; }
      else { far = atomicExch(&s.mark_far[v], s.far_epoch) != s.far_epoch; }
    }
    if (near && s.out_offsets) deg_acc += (unsigned long long)(eoff(s.out_offsets, v + 1) - eoff(s.out_offsets, v));
    wq_near.push(near, (int32_t)v);
    wq_far.push(far, (int32_t)v);
  }
  wq_near.flush();
  wq_far.flush();
  for (int o = 32; o > 0; o >>= 1) deg_acc += __shfl_xor(deg_acc, o);
  if (lane == 0 && deg_acc) atomicAdd(&cnt_replica(s.cnt)->out_edges, deg_acc);
}
template <typename WT>
__global__ void __launch_bounds__(TV_BLOCK) k_sssp_pull_big(int32_t const* bigq, int32_t const* in_offsets, int32_t const* in_indices, sssp_state<WT> s, uint32_t const* fbits)
{
  __shared__ wave_queue_storage<3> wqs;
  wqs.init();
  sssp_pull_fn<WT> f{sssp_relax<WT>{s, wave_queue(wqs, 0, s.q_next, &s.cnt->n_next), wave_queue(wqs, 1, s.far, &s.cnt->n_far), wave_queue(wqs, 2, s.q_set, &s.cnt->n_set)}, fbits};
  expand_big(bigq, in_offsets, in_indices, s.cnt, f);
  f.flush();
}

template <typename WT, bool PK = false>
__global__ void __launch_bounds__(TV_BLOCK) k_sssp_expand(int32_t const* q, int64_t n, int32_t const* offsets, int32_t const* row_end,
                                                          int32_t const* indices, int32_t* bigq, sssp_state<WT> s, int32_t big_deg)
{
  __shared__ wave_queue_storage<3> wqs;
  wqs.init();
  sssp_relax<WT, PK> f{s, wave_queue(wqs, 0, s.q_next, &s.cnt->n_next), wave_queue(wqs, 1, s.far, &s.cnt->n_far), wave_queue(wqs, 2, s.q_set, &s.cnt->n_set)};
#ifdef CGA_SSSP_NO_MLP
  expand_frontier(q, n, offsets, indices, bigq, s.cnt, keep_all{}, f, big_deg, row_end);
#else
  expand_frontier_mlp(q, n, offsets, indices, bigq, s.cnt, keep_all{}, f, big_deg, row_end);
#endif
  f.flush();
}
template <typename WT, bool PK = false>
__global__ void __launch_bounds__(TV_BLOCK) k_sssp_expand_big(int32_t const* bigq, int32_t const* offsets, int32_t const* row_end,
                                                              int32_t const* indices, sssp_state<WT> s)
{
  __shared__ wave_queue_storage<3> wqs;
  wqs.init();
  sssp_relax<WT, PK> f{s, wave_queue(wqs, 0, s.q_next, &s.cnt->n_next), wave_queue(wqs, 1, s.far, &s.cnt->n_far), wave_queue(wqs, 2, s.q_set, &s.cnt->n_set)};
#ifdef CGA_SSSP_NO_MLP
  expand_big(bigq, offsets, indices, s.cnt, f, row_end);
#else
  expand_big_mlp(bigq, offsets, indices, s.cnt, f, row_end);
#endif
  f.flush();
}

// far pile -> (near frontier | far pile'): d < lower: settled meanwhile, drop; d < upper: near; else keep
template <typename WT, bool PK = false>  // PK: `dist` points at the packed (distance, parent) words
__global__ void __launch_bounds__(TV_BLOCK) k_sssp_split(int32_t const* far_in, int64_t n, typename dist_bits<WT>::type const* dist, WT lower,
                                                         WT upper, int32_t* near_out, int32_t* far_out, uint32_t* mark_near,
                                                         uint32_t* mark_far, uint32_t round, uint32_t new_epoch, counters_t* cnt,
                                                         int32_t* set_out, uint32_t* mark_set, uint32_t set_epoch, int32_t const* out_offsets)
{
  using B        = dist_bits<WT>;
  int const lane = threadIdx.x & 63;
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t n_pad  = (n + 63) & ~(int64_t)63;
  unsigned long long kept_min = ~0ull;  // smallest distance bits this lane kept in the far pile
  unsigned long long deg_acc = 0;       // out-degrees of the vertices that enter the near frontier (the next round's edge count)
  for (; i < n_pad; i += stride) {
    bool near = false, keep = false;
    int32_t v = 0;
    if (i < n) {
      v    = far_in[i];
      WT d;
      if constexpr (PK) d = B::from(pk_dist_bits(reinterpret_cast<unsigned long long const*>(dist), v));
      else d = B::from(dist[v]);
      if (d >= lower) {
        if (d < upper) near = atomicExch(&mark_near[v], round) != round;
        else {
          keep = atomicExch(&mark_far[v], new_epoch) != new_epoch;
          if (keep) kept_min = min(kept_min, (unsigned long long)B::to(d));
        }
      }
    }
    if (near && out_offsets) deg_acc += (unsigned long long)(eoff(out_offsets, v + 1) - eoff(out_offsets, v));
    wave_push(near, v, near_out, &cnt->n_next, lane);
    wave_push(keep, v, far_out, &cnt->n_far, lane);
    if (set_out) {  // (uniform) the bucket's members, once each
      bool const fresh = near && atomicExch(&mark_set[v], set_epoch) != set_epoch;
      wave_push(fresh, v, set_out, &cnt->n_set, lane);
    }
  }
  for (int o = 32; o > 0; o >>= 1) deg_acc += __shfl_xor(deg_acc, o);
  if (lane == 0 && deg_acc) atomicAdd(&cnt_replica(cnt)->out_edges, deg_acc);
  // one atomicMin per wavefront (per kept vertex they would all hit the same word)
  for (int o = 32; o > 0; o >>= 1) kept_min = min(kept_min, (unsigned long long)__shfl_xor(kept_min, o));
  if (lane == 0 && kept_min != ~0ull) {
    if constexpr (sizeof(WT) == 4) atomicMin(&cnt->far_min_bits_lo, (uint32_t)kept_min);
    else atomicMin(&cnt->far_min_bits64, kept_min);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// SSSP, round 3: the near set of a bucket is kept in SSSP_K distance-ordered sub-queues (delta-stepping INSIDE the near-far
// window; sssp_impl.cuh:376-561 has one near bucket).  With one near queue every vertex of [lower, upper) is expanded as soon as
// it is reached and again whenever its distance improves inside the window -- 2.2-2.3 relaxations per edge at RMAT-24 with
// weights 1..255, a Bellman-Ford inside the first window.  Here a successful relaxation with nd < upper appends v to sub-queue
// k = floor((nd - lower) / (delta / K)); the sub-queues are drained in order (k == current: the next round of the same sub-queue),
// so a vertex is expanded when the vertices that can still improve it by more than delta / K have been expanded already.  No far
// pile is rescanned between sub-queues (that is what made a narrower delta slower in round 2), stale entries are dropped when they
// are popped: an entry is expanded iff the vertex's CURRENT distance falls into the sub-queue being drained and it has not been
// expanded at that distance (done[v]).  Distances are the same fixed point as before (bit-identical to Dijkstra).
constexpr int SSSP_K = 12;
template <typename WT>
struct sssp_multi_state {
  using bits_t = typename dist_bits<WT>::type;
  bits_t* dist;
  WT const* weights;
  bits_t* done;                 // distance bits at which the vertex was expanded last (unreached pattern = never)
  int32_t* q[SSSP_K];           // sub-queues of the current window (q[k], k > j, receive entries while sub-queue j is drained)
  int32_t* q_same;              // next round of sub-queue j
  int32_t* far;
  uint32_t* mark;               // per vertex: tag of its last insertion into a near queue (dedup within a round / a sub-queue)
  uint32_t* mark_far;
  counters_t* cnt;              // n_next = entries of q_same, n_far = far pile, edges = relaxations
  uint32_t* qn;                 // [SSSP_K] fill of the sub-queues (persist across the rounds of a window)
  uint32_t* qh;                 // [SSSP_K] how many of those entries are vertices below heavy_cut (ids are degree-sorted: the bucket's work)
  WT lower;                     // distances below are settled
  WT ub[SSSP_K];                // absolute, non-decreasing upper bounds: sub-queue k holds [k ? ub[k - 1] : lower, ub[k]); ub[SSSP_K - 1] = the window's end
  WT cutoff;
  int32_t heavy_cut;
  __host__ __device__ WT upper() const { return ub[SSSP_K - 1]; }
  int j;
  uint32_t round_tag, far_epoch;  // tag of an insertion into q_same (one per round) / epoch of the far pile
  uint32_t tag[SSSP_K];           // tag of an insertion into sub-queue k: one per INCARNATION of the sub-queue (a new one whenever its range is
                                  // redefined), so a vertex enters an incarnation at most once and a queue never holds more than V entries
};
template <typename WT>
__device__ __forceinline__ int sssp_sub_of(WT d, sssp_multi_state<WT> const& s)
{  // first k with d < ub[k] (callers have checked d < ub[SSSP_K - 1])
  int k = 0;
#pragma unroll
  for (int i = 0; i < SSSP_K - 1; ++i) k += d >= s.ub[i] ? 1 : 0;
  return k;
}
// per-wavefront LDS staging for SSSP_K + 2 output queues (the wave_queue scheme with the queue picked per call, wave-uniform)
struct multi_queue_storage {
  int32_t buf[SSSP_K + 2][TV_WAVES][WQ_CAP / 4];
  uint32_t fill[SSSP_K + 2][TV_WAVES];
  uint32_t heavy[SSSP_K][TV_WAVES];  // staged entries below heavy_cut, per sub-queue
  int32_t* q[SSSP_K + 2];
  uint32_t* counter[SSSP_K + 2];
  uint32_t* hcounter[SSSP_K];
  uint32_t tag[SSSP_K];
  int32_t heavy_cut;
};
constexpr int MQ_CAP = WQ_CAP / 4;
__device__ __forceinline__ void mq_push(multi_queue_storage& st, int k, bool flag, int32_t value)
{  // k wave-uniform
  uint64_t const m = __ballot(flag);
  if (m == 0) return;
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int32_t* const buf   = st.buf[k][wave];
  uint32_t* const fill = &st.fill[k][wave];
  uint32_t const c     = (uint32_t)__popcll(m);
  int const leader     = __ffsll((unsigned long long)m) - 1;
  uint32_t const rank  = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  uint32_t base        = 0;
  if (k < SSSP_K) {  // (k is wave-uniform)
    uint32_t const hc = (uint32_t)__popcll(__ballot(flag && value < st.heavy_cut));
    if (lane == leader && hc) st.heavy[k][wave] += hc;  // this wavefront's own word
  }
  if (lane == leader) base = atomicAdd(fill, c);
  base = __shfl(base, leader);
  if (base + c <= (uint32_t)MQ_CAP) {
    if (flag) buf[base + rank] = value;
    return;
  }
  uint32_t g = 0;
  if (lane == leader) { g = atomicAdd(st.counter[k], base + c); *fill = 0; }
  g = __shfl(g, leader);
  uint64_t const act = __ballot(true);
  uint32_t const na = (uint32_t)__popcll(act), ar = (uint32_t)__popcll(act & ((1ull << lane) - 1ull));
  int32_t* const q = st.q[k];
  for (uint32_t i = ar; i < base; i += na) q[g + i] = buf[i];
  if (flag) q[g + base + rank] = value;
}
__device__ __forceinline__ void mq_flush(multi_queue_storage& st)
{
  __builtin_amdgcn_wave_barrier();
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < SSSP_K) {
    uint32_t const hc = st.heavy[lane][wave];
    if (hc) { atomicAdd(st.hcounter[lane], hc); st.heavy[lane][wave] = 0; }
  }
  for (int k = 0; k < SSSP_K + 2; ++k) {
    uint32_t const n = st.fill[k][wave];
    if (n == 0) continue;
    uint32_t g = 0;
    if (lane == 0) g = atomicAdd(st.counter[k], n);
    g = __shfl(g, 0);
    int32_t* const q = st.q[k];
    for (uint32_t i = lane; i < n; i += 64) q[g + i] = st.buf[k][wave][i];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) st.fill[k][wave] = 0;
  }
}
template <typename WT>
__device__ __forceinline__ void mq_init(multi_queue_storage& st, sssp_multi_state<WT> const& s)
{
  if (threadIdx.x < (SSSP_K + 2) * TV_WAVES) (&st.fill[0][0])[threadIdx.x] = 0;
  if (threadIdx.x < SSSP_K * TV_WAVES) (&st.heavy[0][0])[threadIdx.x] = 0;
  if (threadIdx.x == 0) st.heavy_cut = s.heavy_cut;
  if (threadIdx.x < SSSP_K) { st.q[threadIdx.x] = s.q[threadIdx.x]; st.counter[threadIdx.x] = &s.qn[threadIdx.x]; st.hcounter[threadIdx.x] = &s.qh[threadIdx.x]; st.tag[threadIdx.x] = s.tag[threadIdx.x]; }
  if (threadIdx.x == SSSP_K) { st.q[SSSP_K] = s.q_same; st.counter[SSSP_K] = &s.cnt->n_next; }
  if (threadIdx.x == SSSP_K + 1) { st.q[SSSP_K + 1] = s.far; st.counter[SSSP_K + 1] = &s.cnt->n_far; }
  __syncthreads();
}
// (the sub-queue pointers are read from `src` -- the kernel-argument copy when `s` is a modified local one: a runtime-indexed read of a
// local struct would go through scratch memory)
template <typename WT>
__device__ __forceinline__ void mq_init_from(multi_queue_storage& st, sssp_multi_state<WT> const& src, sssp_multi_state<WT> const& s)
{
  if (threadIdx.x < (SSSP_K + 2) * TV_WAVES) (&st.fill[0][0])[threadIdx.x] = 0;
  if (threadIdx.x < SSSP_K * TV_WAVES) (&st.heavy[0][0])[threadIdx.x] = 0;
  if (threadIdx.x == 0) st.heavy_cut = s.heavy_cut;
  if (threadIdx.x < SSSP_K) { st.q[threadIdx.x] = src.q[threadIdx.x]; st.counter[threadIdx.x] = &src.qn[threadIdx.x]; st.hcounter[threadIdx.x] = &src.qh[threadIdx.x]; st.tag[threadIdx.x] = src.tag[threadIdx.x]; }
  if (threadIdx.x == SSSP_K) { st.q[SSSP_K] = s.q_same; st.counter[SSSP_K] = &s.cnt->n_next; }
  if (threadIdx.x == SSSP_K + 1) { st.q[SSSP_K + 1] = s.far; st.counter[SSSP_K + 1] = &s.cnt->n_far; }
  __syncthreads();
}
template <typename WT>
struct sssp_relax_multi {
  sssp_multi_state<WT> s;
  multi_queue_storage* st;
  __device__ __forceinline__ void operator()(int32_t u, int32_t v, eoff_t p)
  {
    using B  = dist_bits<WT>;
    WT const nd = B::from(s.dist[u]) + s.weights[p];
    bool near = false, far = false;
    int k = 0;
    if (nd < s.cutoff && nd < B::from(__hip_atomic_load(&s.dist[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
      auto old = atomicMin(&s.dist[v], B::to(nd));
      if (B::to(nd) < old) {
        if (nd < s.upper()) {
          k = sssp_sub_of<WT>(nd, s);
          if (k < s.j) k = s.j;  // (cannot happen for a monotone sub_of; keeps the queue order safe)
          uint32_t const tag = k == s.j ? s.round_tag : st->tag[k];
          near = atomicExch(&s.mark[v], tag) != tag;
        } else {
          far = atomicExch(&s.mark_far[v], s.far_epoch) != s.far_epoch;
        }
      }
    }
    uint64_t m = __ballot(near);
    while (m) {  // one staged append per distinct sub-queue among the lanes
      int const l  = __ffsll((unsigned long long)m) - 1;
      int const k0 = __shfl(k, l);
      bool const mine = near && k == k0;
      mq_push(*st, k0 == s.j ? SSSP_K : k0, mine, v);
      m &= ~__ballot(mine);
    }
    mq_push(*st, SSSP_K + 1, far, v);
  }
};
// pop filter: expand u iff its current distance lies in the sub-queue being drained and it was not expanded at that distance
template <typename WT>
struct sssp_pop {
  typename dist_bits<WT>::type const* dist;
  typename dist_bits<WT>::type* done;
  WT lb, ub;  // the range of the sub-queue being drained
  __device__ __forceinline__ bool operator()(int32_t u) const
  {
    using B = dist_bits<WT>;
    auto const b = dist[u];
    WT const d   = B::from(b);
    if (!(d >= lb && d < ub)) return false;
    return atomicExch(&done[u], b) != b;
  }
};
template <typename WT>
__device__ __forceinline__ sssp_pop<WT> sssp_pop_of(sssp_multi_state<WT> const& s)
{
  WT lb = s.lower, ub = s.ub[0];
#pragma unroll
  for (int k = 1; k < SSSP_K; ++k)
    if (k == s.j) { lb = s.ub[k - 1]; ub = s.ub[k]; }
  return sssp_pop<WT>{s.dist, s.done, lb, ub};
}
template <typename WT>
__global__ void __launch_bounds__(TV_BLOCK) k_sssp_expand_multi(int32_t const* q, int64_t n, int32_t const* offsets, int32_t const* indices, int32_t* bigq,
                                                                sssp_multi_state<WT> s, int32_t big_deg)
{
  __shared__ multi_queue_storage st;
  mq_init<WT>(st, s);
  sssp_relax_multi<WT> f{s, &st};
  sssp_pop<WT> keep = sssp_pop_of<WT>(s);
  expand_frontier(q, n, offsets, indices, bigq, s.cnt, keep, f, big_deg);
  mq_flush(st);
}
template <typename WT>
__global__ void __launch_bounds__(TV_BLOCK) k_sssp_expand_big_multi(int32_t const* bigq, int32_t const* offsets, int32_t const* indices, sssp_multi_state<WT> s)
{
  __shared__ multi_queue_storage st;
  mq_init<WT>(st, s);
  sssp_relax_multi<WT> f{s, &st};
  expand_big(bigq, offsets, indices, s.cnt, f);
  mq_flush(st);
}
// ---- device-driven rounds (CUGRAPH_AMD_SSSP_MODE=dev): the host enqueues a BATCH of rounds -- expand, deferred rows, control -- and
// synchronises once per batch; which queue a round drains, how long it is and its dedup tag come from a control block in device
// memory that a one-wavefront control kernel advances after every round (next round of the same sub-queue while it yields entries,
// then the next non-empty sub-queue; `done` once the window is exhausted: the remaining kernels of the batch return at once).  A
// host-driven round costs ~100 us of launches, counter upload and read-back whatever it relaxes; here it costs three back-to-back
// launches.  The control block lives in the padding of counters_t (the n_set line, unused by this path): one read-back brings
// the cursors and the control state.
struct sssp_ctl_t {
  unsigned long long relaxed;  // edges inspected by the rounds of this batch sequence
  uint32_t done;
  uint32_t j;              // sub-queue being drained
  uint32_t n_front;        // entries of the current frontier
  uint32_t front_is_same;  // 0: the frontier is sub-queue j itself, 1: q_same[cur] (a later round of sub-queue j)
  uint32_t cur;
  uint32_t round;          // dedup tag of the running round
  uint32_t rounds;         // rounds run
};
static_assert(sizeof(sssp_ctl_t) <= 14 * 4, "sssp_ctl_t lives in counters_t::padl2[1..14]");
__host__ __device__ inline sssp_ctl_t* sssp_ctl_of(counters_t* cnt) { return reinterpret_cast<sssp_ctl_t*>(&cnt->padl2[1]); }

template <typename WT>
struct sssp_dev_args {
  sssp_multi_state<WT> s;  // the fields of the window; j / round_tag / q_same are taken from the control block by every kernel
  int32_t* qsame[2];
  int32_t narrow_limit;    // frontiers shorter than this defer every row a wavefront would walk alone (big_deg_for)
  int kk;                  // sub-queues in use
};
template <typename WT>
__device__ __forceinline__ bool sssp_dev_round(sssp_dev_args<WT> const& a, sssp_multi_state<WT>& s, int32_t const*& front, int64_t& n)
{
  sssp_ctl_t const c = *sssp_ctl_of(a.s.cnt);
  if (c.done) return false;
  s           = a.s;
  s.j         = (int)c.j;
  s.round_tag = c.round;
  s.q_same    = a.qsame[(c.cur ^ 1u) & 1u];
  int32_t const* f = a.qsame[c.cur & 1u];
  if (!c.front_is_same) {
#pragma unroll
    for (int k = 0; k < SSSP_K; ++k)
      if (k == (int)c.j) f = a.s.q[k];
  }
  front = f;
  n     = (int64_t)c.n_front;
  return true;
}
template <typename WT>
__global__ void __launch_bounds__(TV_BLOCK) k_sssp_expand_dev(int32_t const* offsets, int32_t const* indices, int32_t* bigq, sssp_dev_args<WT> a)
{
  __shared__ multi_queue_storage st;
  sssp_multi_state<WT> s;
  int32_t const* front;
  int64_t n;
  if (!sssp_dev_round<WT>(a, s, front, n)) return;
  mq_init_from<WT>(st, a.s, s);
  sssp_relax_multi<WT> f{s, &st};
  sssp_pop<WT> keep = sssp_pop_of<WT>(s);
  expand_frontier(front, n, offsets, indices, bigq, s.cnt, keep, f, n < (int64_t)a.narrow_limit ? BIG_DEG_NARROW : BIG_DEG);
  mq_flush(st);
}
template <typename WT>
__global__ void __launch_bounds__(TV_BLOCK) k_sssp_expand_big_dev(int32_t const* bigq, int32_t const* offsets, int32_t const* indices, sssp_dev_args<WT> a)
{
  __shared__ multi_queue_storage st;
  sssp_multi_state<WT> s;
  int32_t const* front;
  int64_t n;
  if (!sssp_dev_round<WT>(a, s, front, n)) return;
  if (s.cnt->n_big == 0) return;
  mq_init_from<WT>(st, a.s, s);
  sssp_relax_multi<WT> f{s, &st};
  expand_big(bigq, offsets, indices, s.cnt, f);
  mq_flush(st);
}
// one wavefront, after the two kernels of a round
__global__ void k_sssp_ctl(counters_t* cnt, int kk)
{
  sssp_ctl_t* const ctl = sssp_ctl_of(cnt);
  if (ctl->done) return;
  int const lane = threadIdx.x & 63;
  unsigned long long e = 0;
  if (lane < CNT_REPLICAS) { e = cnt->rep[lane].edges; cnt->rep[lane].edges = 0; }
  for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
  if (lane != 0) return;
  sssp_ctl_t c = *ctl;
  c.relaxed += e + cnt->edges;
  cnt->edges = 0;
  c.rounds += 1;
  c.round += 1;
  uint32_t const n_next = cnt->n_next;
  cnt->n_next = 0;
  cnt->n_big  = 0;
  if (n_next > 0) {
    c.front_is_same = 1;
    c.cur ^= 1u;
    c.n_front = n_next;
  } else {
    uint32_t j = c.j + 1;  // sub-queue c.j is exhausted (entries for it went to q_same while it was drained)
    while (j < (uint32_t)kk && cnt->padl0[j] == 0) ++j;
    if (j >= (uint32_t)kk) {
      c.done = 1;
    } else {
      c.j             = j;
      c.front_is_same = 0;
      c.n_front       = cnt->padl0[j];
      cnt->padl0[j]   = 0;
    }
  }
  *ctl = c;
}

// far pile -> the sub-queues of the new window [lower, upper) | far pile'
template <typename WT>
__global__ void __launch_bounds__(TV_BLOCK) k_sssp_split_multi(int32_t const* far_in, int64_t n, sssp_multi_state<WT> s, int32_t* far_out, uint32_t new_epoch)
{
  using B = dist_bits<WT>;
  __shared__ multi_queue_storage st;
  mq_init<WT>(st, s);
  if (threadIdx.x == 0) st.q[SSSP_K + 1] = far_out;  // the kept entries go to the other far buffer
  __syncthreads();
  int const lane = threadIdx.x & 63;
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t n_pad  = (n + 63) & ~(int64_t)63;
  unsigned long long kept_min = ~0ull;
  for (; i < n_pad; i += stride) {
    bool near = false, keep = false;
    int32_t v = 0;
    int k = 0;
    if (i < n) {
      v    = far_in[i];
      WT d = B::from(s.dist[v]);
      if (d >= s.lower) {
        if (d < s.upper()) {
          k = sssp_sub_of<WT>(d, s);
          uint32_t const tag = st.tag[k];
          near = atomicExch(&s.mark[v], tag) != tag;
        } else {
          keep = atomicExch(&s.mark_far[v], new_epoch) != new_epoch;
          if (keep) kept_min = min(kept_min, (unsigned long long)B::to(d));
        }
      }
    }
    uint64_t m = __ballot(near);
    while (m) {
      int const l  = __ffsll((unsigned long long)m) - 1;
      int const k0 = __shfl(k, l);
      bool const mine = near && k == k0;
      mq_push(st, k0, mine, v);
      m &= ~__ballot(mine);
    }
    mq_push(st, SSSP_K + 1, keep, v);
  }
  mq_flush(st);
  for (int o = 32; o > 0; o >>= 1) kept_min = min(kept_min, (unsigned long long)__shfl_xor(kept_min, o));
  if (lane == 0 && kept_min != ~0ull) {
    if constexpr (sizeof(WT) == 4) atomicMin(&s.cnt->far_min_bits_lo, (uint32_t)kept_min);
    else atomicMin(&s.cnt->far_min_bits64, kept_min);
  }
}

// canonical parents: pred[v] = min EXTERNAL id u with d[u] + w(u,v) == d[v]
template <typename WT>
struct sssp_parent {
  typename dist_bits<WT>::type const* dist;
  WT const* weights;
  int32_t* pred;  // INT32_MAX = none
  int32_t const* labels;
  int32_t source;
  __device__ __forceinline__ void operator()(int32_t u, int32_t v, eoff_t p) const
  {
    using B = dist_bits<WT>;
    if (v != source && B::to(B::from(dist[u]) + weights[p]) == dist[v]) atomicMin(&pred[v], labels ? labels[u] : u);
  }
  // the phased form (expand_*_mlp: EX_U edges in flight per lane; the sweep is one random read of d[v] per settled edge, nearly all of
  // which fail the test -- the kind of round that gained most from several loads in flight, section 3.4 of DESIGN.md)
  struct cand_t { int32_t label; bool pass; };
  using tok_t  = int32_t;
  using tok2_t = int32_t;
  __device__ __forceinline__ cand_t pre(int32_t u, int32_t v, eoff_t p) const
  {
    using B = dist_bits<WT>;
    int32_t const uu = u < 0 ? 0 : u, vv = v < 0 ? 0 : v;
    bool const tight = (v >= 0) & (v != source) & (B::to(B::from(dist[uu]) + weights[p]) == dist[vv]);
    return cand_t{labels ? labels[uu] : uu, tight};
  }
  __device__ __forceinline__ tok_t mid(int32_t v, cand_t c) const
  {
    if (c.pass && c.label < pred[v]) atomicMin(&pred[v], c.label);  // (plain pre-test: pred only ever decreases)
    return 0;
  }
  __device__ __forceinline__ tok2_t mid2(int32_t, cand_t, tok_t) const { return 0; }
  __device__ __forceinline__ void post(int32_t, int32_t, cand_t, tok_t, tok2_t) const {}
};
template <typename WT>
struct keep_reached {
  typename dist_bits<WT>::type const* dist;
  typename dist_bits<WT>::type unreached;
  __device__ __forceinline__ bool operator()(int32_t u) const { return dist[u] != unreached; }
};
template <typename WT>
__global__ void __launch_bounds__(TV_BLOCK) k_sssp_parents(int64_t nv, int32_t const* offsets, int32_t const* indices, int32_t* bigq,
                                                           counters_t* cnt, sssp_parent<WT> f, keep_reached<WT> keep)
{
  expand_frontier_mlp((int32_t const*)nullptr, nv, offsets, indices, bigq, cnt, keep, f);
}
template <typename WT>
__global__ void __launch_bounds__(TV_BLOCK) k_sssp_parents_big(int32_t const* bigq, int32_t const* offsets, int32_t const* indices,
                                                               counters_t* cnt, sssp_parent<WT> f)
{
  expand_big_mlp(bigq, offsets, indices, cnt, f);
}

// keys for ordering a near frontier by tentative distance -- 256 bands of the current window -- so that the vertices closest to the source
// are expanded first inside a round (run_sssp: the hub round)
template <typename WT, bool PK>
__global__ void k_sssp_band_keys(int32_t const* front, int64_t n, typename dist_bits<WT>::type const* dist, WT lower, WT inv_band, uint64_t* keys, uint32_t* vals)
{
  using B = dist_bits<WT>;
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    int32_t const v = front[i];
    WT d;
    if constexpr (PK) d = B::from(pk_dist_bits(reinterpret_cast<unsigned long long const*>(dist), v));
    else d = B::from(dist[v]);
    WT const b = (d - lower) * inv_band;
    keys[i] = (uint64_t)(b < WT(0) ? 0 : (b > WT(255) ? 255 : (int)b));
    vals[i] = (uint32_t)v;
  }
}

// packed (distance, parent) words -> the two result columns; parent INT32_MAX (none: unreached, or the source) -> -1
__global__ void k_sssp_unpack(unsigned long long const* pk, int64_t n, uint32_t* dist_bits_out, int32_t* pred)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned long long const w = pk[i];
    dist_bits_out[i] = (uint32_t)(w >> 32);
    int32_t const p  = (int32_t)(uint32_t)w;
    pred[i]          = p == INT32_MAX ? -1 : p;
  }
}

template <typename WT>
__global__ void k_sum_weights(WT const* w, int64_t n, double* out)  // out[0] += sum; out[1] = number of weights that differ from w[0]
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double s       = 0, differ = 0;
  WT const w0    = w[0];
  for (; i < n; i += stride) { WT const x = w[i]; s += (double)x; differ += x != w0 ? 1.0 : 0.0; }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); differ += __shfl_xor(differ, o); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(out, s); if (differ > 0.0) atomicAdd(out + 1, differ); }
}

__global__ void k_fix_pred(int32_t* pred, int64_t n)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride)
    if (pred[i] == INT32_MAX) pred[i] = -1;  // invalid_vertex_id
}

__global__ void k_count_degree_at_least(int32_t const* offsets, int64_t nv, uint32_t deg, unsigned long long* out)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned c     = 0;
  for (; i < nv; i += stride) c += (eoff(offsets, i + 1) - eoff(offsets, i)) >= deg;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

template <typename T>
__global__ void k_fill_t(T* p, int64_t n, T v)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

__global__ void k_bfs_edges_of_reached(int32_t const* dist, int32_t const* out_offsets, int64_t nv, unsigned long long* out)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long c = 0;
  for (; i < nv; i += stride)
    if (dist[i] != INT32_MAX) c += (unsigned long long)(eoff(out_offsets, i + 1) - eoff(out_offsets, i));
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

__global__ void k_count_reached(uint32_t const* vis, int64_t nwords, unsigned long long* out)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned c     = 0;
  for (; i < nwords; i += stride) c += __popc(vis[i]);
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

template <typename B>
__global__ void k_count_reached_dist(B const* dist, int64_t nv, B unreached, unsigned long long* out)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned c     = 0;
  for (; i < nv; i += stride) c += dist[i] != unreached;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

// (row, segment) pairs: at most E / BIG_DEG deferred rows, each with ceil(deg / BIG_SEG) <= deg / BIG_SEG + 1 segments

// ------------------------------------------------------------------------------------------ drivers
paths_result_t* run_bfs(handle_t& h, graph_t& g, device_array_view_t const* sources, bool direction_optimizing, size_t depth_limit,
                        bool compute_predecessors)
{
  HIP_TRY(hipSetDevice(h.device));
  static bool const trace = getenv("CUGRAPH_AMD_BFS_TRACE") != nullptr;
  auto const t_trace0 = std::chrono::steady_clock::now();
  auto mark = [&](char const* what, long long a0 = 0, long long a1 = 0) {
    if (trace) {
      h.sync();
      fprintf(stderr, "[bfs] %8.1f us  %s %lld %lld\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_trace0).count(), what, a0, a1);
      fflush(stderr);
    }
  };
  CGA_EXPECTS(sources != nullptr, CUGRAPH_INVALID_INPUT, "sources is NULL");
  vertex_column_in c_sources;  // INT64 / sparse external ids: compact int32 ids from here on (outer_ids.hip)
  sources = c_sources.get(h, g, sources, "sources");
  if (direction_optimizing)  // bfs_impl.cuh:202-204
    CGA_EXPECTS(g.props.is_symmetric == TRUE, CUGRAPH_INVALID_INPUT,
                "Invalid input argument: input graph should be symmetric for direction optimizing BFS.");
  ensure_orientation(h, g, false);  // top-down pushes over CSR
  orientation_t const& o = g.csr;
  int64_t const nv = g.nv, ns = (int64_t)sources->size;
  size_t const n1  = (size_t)(nv > 0 ? nv : 1);

  // Bottom-up levels need the in-edges.  A symmetric graph's CSR is its CSC; otherwise the CSC is used when it exists
  // already (e.g. after a PageRank call) and built from the second BFS on (it costs about as much as three traversals).
  // The direction never changes distances, so this is independent of the API's direction_optimizing flag.
  char const* env = getenv("CUGRAPH_AMD_BFS");  // "topdown" pins the push-only path (testing / profiling)
  bool const pin_topdown = env && std::string(env) == "topdown";
  ++g.bfs_calls;
  orientation_t const* in = nullptr;
  if (!pin_topdown) {
    if (g.props.is_symmetric == TRUE) in = &g.csr;
    else if (g.csc.built || g.bfs_calls >= 2 || (env && std::string(env) == "bottomup")) { ensure_orientation(h, g, true); in = &g.csc; }
  }
  int32_t const* in_off  = in ? in->offsets.data() : o.offsets.data();
  int32_t const* in_idx  = in ? in->indices.data() : nullptr;
  int32_t const* out_off = o.offsets.data();

  dvec<int32_t> src(ns > 0 ? ns : 1);
  if (ns > 0) HIP_TRY(hipMemcpyAsync(src.data(), sources->data, ns * 4, hipMemcpyDeviceToDevice, h.stream));
  renumber_ext_to_int(h, g, src.data(), ns);  // ids that are not vertices become -1: counted by k_bfs_init_sources (no extra round trip)

  auto ids   = std::make_unique<device_array_t>((size_t)nv, g.vertex_type);
  auto dist  = std::make_unique<device_array_t>((size_t)nv, g.vertex_type);
  auto preds = std::make_unique<device_array_t>(compute_predecessors ? (size_t)nv : 0, g.vertex_type);
  int64_t const nwords = ((nv + 63) / 64) * 2 + 2;  // whole 64-vertex groups
  dvec<uint32_t> vis_prev(nwords), vis_new(nwords), front(in ? nwords : 1), next(in ? nwords : 1);
  dvec<int32_t> qa(n1), qb(n1), bigq(big_queue_entries(g.ne));
  dvec<counters_t> cnt(1);
  int32_t* const pred_p = compute_predecessors ? preds->buf.as<int32_t>() : nullptr;
  int32_t const* labels = g.renumbered ? g.number_map.data() : nullptr;
  // one launch for the whole initial state (it used to be two fills and five memsets: 50 us of a 1.2 ms search at RMAT-24).  With in-edges
  // the bottom-up kernel rewrites whole 64-vertex groups only: the two slack words of front / next must read as "no vertex"
  hipLaunchKernelGGL(k_bfs_init_state, grid_for(std::max<int64_t>(nv / 4, nwords), kBlock, 4096), kBlock, 0, h.stream, dist->buf.as<int32_t>(), pred_p, nv, vis_prev.data(),
                     vis_new.data(), in ? front.data() : (uint32_t*)nullptr, in ? next.data() : (uint32_t*)nullptr, nwords, cnt.data());
  if (ns > 0)
    hipLaunchKernelGGL(k_bfs_init_sources, grid_for(ns, kBlock), kBlock, 0, h.stream, (int32_t const*)src.data(), ns, dist->buf.as<int32_t>(),
                       vis_prev.data(), vis_new.data(), qa.data(), cnt.data(), out_off, in_off);
  mark("init", nv, ns);
  counters_t c;
  h.read_back(&c, cnt.data(), 1);
  c.fold();
  CGA_EXPECTS(c.n_big == 0, CUGRAPH_INVALID_INPUT, "Found invalid vertex in the input sources");  // bfs.cpp:106-119
  int64_t n_cur  = c.n_next;
  int32_t* q_cur = qa.data();
  int32_t* q_nxt = qb.data();
  uint64_t depth = 0, edges = 0, levels = 0, bu_levels = 0;
  // Beamer's heuristic: go bottom-up when the frontier's out-edges exceed 1/alpha of the unvisited vertices' in-edges,
  // come back when the frontier has shrunk below V / beta.  alpha = 60 (Beamer: 14): a bottom-up level costs 0.2-0.3 ms at RMAT-24
  // whatever the frontier, a top-down level ~25-70 G edges/s -- the level of the ~10^3 hubs right after the source (10-20 M
  // out-edges, 0.7 ms top-down) is cheaper bottom-up (64 roots: mean 1.47 -> 1.38 ms, max 1.93 -> 1.61 ms)
  char const* env_alpha = getenv("CUGRAPH_AMD_BFS_ALPHA");
  char const* env_beta  = getenv("CUGRAPH_AMD_BFS_BETA");
  double const alpha = env_alpha ? atof(env_alpha) : 60.0, beta = env_beta ? atof(env_beta) : 24.0;
  uint64_t frontier_out = c.out_edges;            // out-edges of the current frontier
  uint64_t reached_total = c.n_next, edges_of_reached = c.out_edges;
  uint64_t unvisited_in = (uint64_t)g.ne - c.in_edges;
  bool bottom_up = false, front_is_bitmap = false;
  // depth_limit is compared after incrementing (bfs_impl.cuh:867-868)
  uint64_t const limit = depth_limit > (size_t)INT32_MAX ? (uint64_t)INT32_MAX : (uint64_t)depth_limit;
  bool const bu_profile = getenv("CUGRAPH_AMD_BFS_PROFILE") != nullptr;
  // edges per deferred work unit of a NARROW frontier (CUGRAPH_AMD_BFS_NARROW_SEG: 64 ... 4096, power of two not required)
  int32_t const narrow_seg = getenv("CUGRAPH_AMD_BFS_NARROW_SEG") ? std::max(64, std::min(4096, atoi(getenv("CUGRAPH_AMD_BFS_NARROW_SEG")))) : BIG_SEG_NARROW;
  // OPT-IN (CUGRAPH_AMD_BFS_PULL_PARENTS=1), measured at RMAT-24 (32 roots, profiles/r5d_bfs_ab.txt): 1.74 ms against 1.37 ms with the
  // atomicMin in the push.  The levels that run top-down are the ones whose discoveries are HUBS (one frontier vertex discovering 1 141 vertices
  // of 10^4-10^5 in-edges each): a pull scans half of every such row to find the one frontier member, the push reads the frontier's out-edges once.
  bool const pull_parents_on = getenv("CUGRAPH_AMD_BFS_PULL_PARENTS") != nullptr && atoi(getenv("CUGRAPH_AMD_BFS_PULL_PARENTS")) != 0;
  char const* env_bug = getenv("CUGRAPH_AMD_BU_GRID");  // workgroups per CU of the bottom-up kernel (experiments)
  int const bu_grid = (int)std::max<int64_t>(1, std::min<int64_t>(((nv + 63) / 64 + TV_WAVES * BU_GROUPS - 1) / (TV_WAVES * BU_GROUPS),
                                                                  (int64_t)h.num_cus * (env_bug ? atoi(env_bug) : 16)));
  while (n_cur > 0) {
    if (in) {
      if (!bottom_up) bottom_up = (double)frontier_out > (double)unvisited_in / alpha && n_cur > 1024;
      else bottom_up = !((double)n_cur < (double)nv / beta);
      if (env && std::string(env) == "bottomup") bottom_up = true;
    }
    HIP_TRY(hipMemsetAsync(cnt.data(), 0, sizeof(counters_t), h.stream));
    if (bottom_up) {
      if (!front_is_bitmap)  // the last level ran top-down (queue): snapshot the visited set as the bitmap to test against
        hipLaunchKernelGGL(k_bfs_front_from_vis, grid_for(nwords, kBlock, 2048), kBlock, 0, h.stream, vis_prev.data(), (uint32_t const*)vis_new.data(),
                           front.data(), nwords);
      {
        timed_launch t(h, "bfs_bottom_up");
        if (bu_profile) {
          dvec<unsigned long long> prof(8);
          HIP_TRY(hipMemsetAsync(prof.data(), 0, 8 * sizeof(unsigned long long), h.stream));
          hipLaunchKernelGGL(k_bfs_bottom_up<true>, bu_grid, TV_BLOCK, 0, h.stream, in_off, in_idx, out_off, nv, vis_new.data(), (uint32_t const*)front.data(),
                             next.data(), dist->buf.as<int32_t>(), pred_p, (int32_t)(depth + 1), cnt.data(), prof.data());
          unsigned long long pr[8];
          h.read_back(pr, (unsigned long long const*)prof.data(), 8);
          double const nw = (double)bu_grid * TV_WAVES;
          fprintf(stderr, "[bfs bottom-up profile] depth %lld: per wavefront (100 MHz ticks) first chunk %.0f, lane tail %.0f, long rows + write %.0f; slowest wavefront %llu "
                          "(long rows %llu); tail steps %.1f, long rows %.2f, long-row steps %.1f per wavefront\n",
                  (long long)depth, pr[0] / nw, pr[1] / nw, pr[2] / nw, pr[6], pr[7], pr[3] / nw, pr[4] / nw, pr[5] / nw);
        } else {
          hipLaunchKernelGGL(k_bfs_bottom_up<false>, bu_grid, TV_BLOCK, 0, h.stream, in_off, in_idx, out_off, nv, vis_new.data(), (uint32_t const*)front.data(),
                             next.data(), dist->buf.as<int32_t>(), pred_p, (int32_t)(depth + 1), cnt.data(), (unsigned long long*)nullptr);
        }
      }
      mark("bottom_up", (long long)depth, n_cur);
      std::swap(front, next);  // `next` of this level is the frontier bitmap of the following one
      front_is_bitmap = true;
      ++bu_levels;
    } else {
      if (front_is_bitmap) {  // the last level ran bottom-up: materialise its discoveries as a queue
        hipLaunchKernelGGL(k_bfs_bitmap_to_queue, grid_for(nwords, TV_BLOCK, 512), TV_BLOCK, 0, h.stream, (uint32_t const*)front.data(), nwords, q_cur,
                           cnt.data());
        HIP_TRY(hipMemsetAsync(cnt.data(), 0, sizeof(counters_t), h.stream));
        HIP_TRY(hipMemcpyAsync(vis_prev.data(), vis_new.data(), nwords * 4, hipMemcpyDeviceToDevice, h.stream));
        front_is_bitmap = false;
        mark("bitmap_to_queue", (long long)depth, n_cur);
      }
      // with in-edges at hand the parents of this level's discoveries are pulled afterwards (k_bfs_pull_parents); the push leaves pred alone
      bool const pull_parents = pred_p != nullptr && in != nullptr && pull_parents_on;
      bfs_state s{dist->buf.as<int32_t>(), pull_parents ? (int32_t*)nullptr : pred_p, vis_prev.data(), vis_new.data(), q_nxt,
                  cnt.data(), out_off, in_off, (int32_t)(depth + 1)};
      {
        timed_launch t(h, "bfs_expand");
        int32_t const seg = big_seg_for_deg(big_deg_for(h, n_cur), narrow_seg);
        hipLaunchKernelGGL(k_bfs_expand, expand_grid(h, n_cur), TV_BLOCK, 0, h.stream, (int32_t const*)q_cur, n_cur, (int32_t const*)o.offsets.data(),
                           (int32_t const*)o.indices.data(), bigq.data(), s, big_deg_for(h, n_cur), seg);
        hipLaunchKernelGGL(k_bfs_expand_big, h.num_cus * 8, TV_BLOCK, 0, h.stream, (int32_t const*)bigq.data(), (int32_t const*)o.offsets.data(),
                           (int32_t const*)o.indices.data(), s, seg);
        if (pull_parents) {  // (vis_prev still is the visited set of the level's start; bigq is free again)
          hipLaunchKernelGGL(k_bfs_pull_parents, h.num_cus * 4, TV_BLOCK, 0, h.stream, (int32_t const*)q_nxt, cnt.data(), in_off, in_idx, (uint32_t const*)vis_prev.data(), pred_p,
                             bigq.data());
          if (in->max_degree > BFS_PULL_LONG)
            hipLaunchKernelGGL(k_bfs_pull_parents_long, h.num_cus * 4, TV_BLOCK, 0, h.stream, (int32_t const*)bigq.data(), (counters_t const*)cnt.data(), in_off, in_idx,
                               (uint32_t const*)vis_prev.data(), pred_p);
        }
      }
      if (!in) HIP_TRY(hipMemcpyAsync(vis_prev.data(), vis_new.data(), nwords * 4, hipMemcpyDeviceToDevice, h.stream));
      mark("top_down", (long long)depth, n_cur);
      std::swap(q_cur, q_nxt);
    }
    h.read_back(&c, cnt.data(), 1);
    c.fold();
    edges += c.edges;
    n_cur        = c.n_next;
    frontier_out = c.out_edges;
    reached_total += c.n_next;
    edges_of_reached += c.out_edges;
    unvisited_in -= std::min<uint64_t>(unvisited_in, c.in_edges);
    if (in && !bottom_up && !front_is_bitmap) {
      // keep vis_prev one level behind only while the next level may need vis_new & ~vis_prev; a following top-down
      // level needs vis_prev = vis_new (done lazily here when the decision is known to be top-down again)
      bool next_bu = ((double)frontier_out > (double)unvisited_in / alpha && n_cur > 1024) || (env && std::string(env) == "bottomup");
      if (!next_bu) HIP_TRY(hipMemcpyAsync(vis_prev.data(), vis_new.data(), nwords * 4, hipMemcpyDeviceToDevice, h.stream));
    }
    ++depth;
    ++levels;
    if (depth >= limit) break;
  }
  // statistics: every discovered vertex was counted (with its out-degree) by the level that found it -- no extra pass
  h.last_stats = cugraph_amd_traversal_stats_t{levels, edges, reached_total, edges_of_reached};
  (void)bu_levels;
  // the vertex column (a copy of the numbering: the result owns its columns) and, bfs.cpp:131-138, the predecessors back in external ids:
  // one kernel (the 64 MB device-to-device copy of the runtime took 80 us at RMAT-24, a streaming kernel takes 30)
  if (nv > 0) hipLaunchKernelGGL(k_bfs_result_columns, grid_for(nv, kBlock, 4096), kBlock, 0, h.stream, (int32_t const*)g.number_map.data(), ids->buf.as<int32_t>(), pred_p, nv, labels);
  mark("finish_pred");
  h.sync();
  auto* r = new paths_result_t{ids.release(), dist.release(), preds.release()};
  outer_replace_ids(h, g, r->vertex_ids);
  outer_replace_dist(h, g, r->distances);  // BFS distances carry the vertex type (bfs.cpp:156-187)
  outer_replace_ids(h, g, r->predecessors);
  return r;
}

// ---- light / heavy copy of the CSR (sssp_lh_t, common.hpp)
template <typename WT>
__global__ void k_lh_keys(int32_t const* offsets, WT const* w, int64_t nv, WT delta, uint64_t* keys, uint32_t* vals)
{  // one wavefront per row: key = row << 1 | heavy, payload = edge position
  int64_t const wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int const lane = threadIdx.x & 63;
  for (int64_t v = wave; v < nv; v += nwaves) {
    int32_t const b = offsets[v], e = offsets[v + 1];
    for (int32_t p = b + lane; p < e; p += 64) { keys[p] = ((uint64_t)v << 1) | (w[p] > delta ? 1u : 0u); vals[p] = (uint32_t)p; }
  }
}
__global__ void k_lh_ends_init(int32_t const* offsets, int64_t nv, int32_t* light_end)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nv; i += stride) light_end[i] = offsets[i + 1];
}
__global__ void k_lh_ends(uint64_t const* keys, int64_t ne, int32_t* light_end)
{  // first position of every (row, heavy) group
  int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for (; p < ne; p += stride) {
    uint64_t const k = keys[p];
    if ((k & 1u) && (p == 0 || keys[p - 1] != k)) light_end[k >> 1] = (int32_t)p;
  }
}
template <typename WT>
std::shared_ptr<sssp_lh_t> build_sssp_lh(handle_t const& h, graph_t const& g, orientation_t const& o, double delta)
{
  build_trace tr(h, "sssp l/h");
  auto lh   = std::make_shared<sssp_lh_t>();
  lh->delta = delta;
  int64_t const nv = g.nv, ne = g.ne;
  lh->indices.resize_discard((size_t)ne + kEdgePad);
  lh->weights.alloc(((size_t)ne + kEdgePad) * sizeof(WT));
  lh->light_end.resize_discard((size_t)std::max<int64_t>(nv, 1));
  HIP_TRY(hipMemsetAsync(lh->indices.data() + ne, 0, kEdgePad * sizeof(int32_t), h.stream));
  HIP_TRY(hipMemsetAsync(static_cast<char*>(lh->weights.ptr) + ne * sizeof(WT), 0, kEdgePad * sizeof(WT), h.stream));
  dvec<uint64_t> keys((size_t)ne), keys_tmp((size_t)ne);
  dvec<uint32_t> vals((size_t)ne), vals_tmp((size_t)ne);
  hipLaunchKernelGGL(k_lh_keys<WT>, grid_for(nv * 16, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)o.offsets.data(), o.weights.as<WT const>(), nv, (WT)delta,
                     keys.data(), vals.data());
  int vb = 1;
  while (vb < 40 && (((uint64_t)std::max<int64_t>(nv - 1, 1)) >> vb) != 0) ++vb;
  radix_sort_u64_u32(h, keys.data(), vals.data(), keys_tmp.data(), vals_tmp.data(), ne, 0, vb + 1);
  gather_b32(h, reinterpret_cast<uint32_t const*>(o.indices.data()), vals.data(), reinterpret_cast<uint32_t*>(lh->indices.data()), ne);
  if (sizeof(WT) == 4) gather_b32(h, o.weights.as<uint32_t const>(), vals.data(), lh->weights.as<uint32_t>(), ne);
  else                 gather_b64(h, o.weights.as<uint64_t const>(), vals.data(), lh->weights.as<uint64_t>(), ne);
  hipLaunchKernelGGL(k_lh_ends_init, grid_for(nv, kBlock, 4096), kBlock, 0, h.stream, (int32_t const*)o.offsets.data(), nv, lh->light_end.data());
  hipLaunchKernelGGL(k_lh_ends, grid_for(ne, kBlock, 8192), kBlock, 0, h.stream, (uint64_t const*)keys.data(), ne, lh->light_end.data());
  h.sync();
  return lh;
}

template <typename WT>
paths_result_t* run_sssp(handle_t& h, graph_t& g, size_t source_ext, double cutoff_d, bool compute_predecessors)
{
  using B      = dist_bits<WT>;
  using bits_t = typename B::type;
  HIP_TRY(hipSetDevice(h.device));
  ensure_orientation(h, g, false);
  orientation_t const& o = g.csr;
  int64_t const nv = g.nv;
  size_t const n1  = (size_t)(nv > 0 ? nv : 1);
  WT const wmax    = std::numeric_limits<WT>::max();
  WT const cutoff  = cutoff_d >= (double)wmax ? wmax : (WT)cutoff_d;

  // source: external id -> internal (sssp.cpp:84-103)
  dvec<int32_t> src(1);
  if (g.outer.active) {  // INT64 / sparse external ids: look the source up in the sorted id list first
    dvec<int64_t> one(1);
    int64_t const sv = (int64_t)source_ext;
    HIP_TRY(hipMemcpyAsync(one.data(), &sv, 8, hipMemcpyHostToDevice, h.stream));
    outer_to_compact(h, g.outer, one.data(), INT64, 1, src.data());
    int32_t c = -1;
    h.read_back(&c, src.data(), 1);
    CGA_EXPECTS(c >= 0, CUGRAPH_INVALID_INPUT, "Invalid input argument: source vertex is not a vertex of the graph.");
    source_ext = (size_t)c;
  }
  CGA_EXPECTS(source_ext <= (size_t)INT32_MAX, CUGRAPH_INVALID_INPUT, "invalid source vertex");
  int32_t s_host = (int32_t)source_ext;
  HIP_TRY(hipMemcpyAsync(src.data(), &s_host, 4, hipMemcpyHostToDevice, h.stream));
  h.sync();
  renumber_ext_to_int(h, g, src.data(), 1);
  h.read_back(&s_host, src.data(), 1);
  CGA_EXPECTS(s_host >= 0 && s_host < nv, CUGRAPH_INVALID_INPUT, "Invalid input argument: source vertex is not a vertex of the graph.");
  int32_t const source = s_host;

  auto ids   = std::make_unique<device_array_t>((size_t)nv, g.vertex_type);
  auto dist  = std::make_unique<device_array_t>((size_t)nv, g.weight_type);
  auto preds = std::make_unique<device_array_t>(compute_predecessors ? (size_t)nv : 0, g.vertex_type);
  bits_t* d  = dist->buf.as<bits_t>();
  WT const* w = o.weights.as<WT const>();
  dvec<int32_t> qa(n1), qb(n1), fa(n1), fb(n1), bigq(big_queue_entries(g.ne));
  dvec<uint32_t> mark_near(n1), mark_far(n1);
  dvec<counters_t> cnt(1);
  dvec<double> wsum(2);

  bits_t unreached_bits;
  {
    WT m = wmax;
    std::memcpy(&unreached_bits, &m, sizeof(WT));
  }
  hipLaunchKernelGGL(k_fill_t<bits_t>, grid_for(nv, kBlock, 4096), kBlock, 0, h.stream, d, nv, unreached_bits);
  HIP_TRY(hipMemsetAsync(mark_near.data(), 0, n1 * 4, h.stream));
  HIP_TRY(hipMemsetAsync(mark_far.data(), 0, n1 * 4, h.stream));
  if (!g.weight_sum_valid) {  // the weights of a graph never change: one pass per graph, not per call
    HIP_TRY(hipMemsetAsync(wsum.data(), 0, 16, h.stream));
    if (g.ne > 0) hipLaunchKernelGGL(k_sum_weights<WT>, grid_for(g.ne, kBlock, 2048), kBlock, 0, h.stream, w, g.ne, wsum.data());
    double ws[2];
    h.read_back(ws, (double const*)wsum.data(), 2);
    g.weight_sum       = ws[0];
    g.weights_uniform  = ws[1] == 0.0;
    g.weight_sum_valid = true;
  }
  double const wsum_h = g.weight_sum;
  // delta = 32 * average weight / average degree (sssp_impl.cuh:233-247)
  double avg_w   = g.ne > 0 ? wsum_h / (double)g.ne : 1.0;
  double avg_deg = nv > 0 ? (double)g.ne / (double)nv : 1.0;
  double delta   = avg_w * 32.0 / std::max(avg_deg, 1.0);
  if (char const* e = getenv("CUGRAPH_AMD_SSSP_DELTA_SCALE")) delta *= atof(e);  // tuning knob (bucket width multiplier)
  if (!(delta > 0.0) || !std::isfinite(delta)) delta = 1.0;
  // Light / heavy buckets (Meyer & Sanders' delta-stepping): with the wide buckets above a vertex is re-expanded every time its
  // distance improves -- 2.4 relaxations per edge at RMAT-24 with weights 1..255.  Narrow buckets cure that only if the edges that
  // cannot land in the current bucket (w > delta: "heavy") are relaxed ONCE, when the bucket closes and its members' distances are
  // final; inside the bucket only the light edges are relaxed.  Needs every row's light edges first: sssp_lh_t, built once per graph.
  char const* env_lh = getenv("CUGRAPH_AMD_SSSP_LH");
  // Measured at RMAT-24, weights 1..255 (16 roots): relaxations per edge 2.25 -> 1.24, rounds 18 -> 46, time 11.5 -> 12.4 ms (delta / 4;
  // delta / 2: 14.9, delta / 8: 15.9): what a round costs is the SUCCESSFUL updates (atomicMin + mark exchange + queue append, and
  // the far pile that every bucket re-splits), not the failed relaxations this scheme removes.  So the path is opt-in
  // (CUGRAPH_AMD_SSSP_LH=1) and covered by test_sssp_light_heavy_buckets_vs_oracle; wide buckets stay the default.
  bool const use_lh  = g.ne > 0 && g.ne <= kMaxSignedEdges && env_lh && atoi(env_lh) != 0;  // (the light / heavy copy keeps signed positions)
  if (use_lh) {
    char const* env_s = getenv("CUGRAPH_AMD_SSSP_LH_SCALE");
    delta *= env_s ? atof(env_s) : 0.25;
    if (!(delta > 0.0) || !std::isfinite(delta)) delta = 1.0;
    orientation_t& ow = g.csr;
    if (!ow.lh || ow.lh->delta != delta) ow.lh = build_sssp_lh<WT>(h, g, o, delta);
  }
  int32_t const* row_beg  = o.offsets.data();
  int32_t const* adj      = use_lh ? g.csr.lh->indices.data() : o.indices.data();
  int32_t const* lend     = use_lh ? g.csr.lh->light_end.data() : nullptr;
  WT const* wrel          = use_lh ? g.csr.lh->weights.template as<WT const>() : w;
  dvec<int32_t> sa(use_lh ? n1 : 1), sb(use_lh ? n1 : 1);
  dvec<uint32_t> mark_set(use_lh ? n1 : 1);
  if (use_lh) HIP_TRY(hipMemsetAsync(mark_set.data(), 0, n1 * 4, h.stream));

  // Opt-in schedules (CUGRAPH_AMD_SSSP_MODE; same fixed point, tested bit for bit against Dijkstra by test_sssp_subqueues_vs_oracle):
  //   multi  distance-ordered sub-queues inside the window (k_sssp_expand_multi): 8 sub-queues 1.64 relaxations per edge instead of 2.2,
  //          49.6 rounds instead of 18.4 -- and 12.7 ms instead of 11.5 at RMAT-24, weights 1..255 (profiles/r3_sssp_subqueues.txt)
  //   dev    the same with the rounds of a window driven from the device (k_sssp_ctl; one host synchronisation per batch of rounds): 12.1 ms
  //   radix  radix-heap sub-queues (below): 0.99-1.4 relaxations per edge, 108 rounds, 21.8 ms
  // What CUGRAPH_AMD_SSSP_TRACE shows (profiles/r3_sssp_rounds.txt): an empty round costs 23 us, not the 100 us assumed in round 2; of the
  // 11 ms of a traversal 9 are FOUR rounds -- the hubs right after the source (66 K vertices, 79 M edges, most relaxations succeed:
  // 24 G edges/s), the sweep over nearly every edge (259 M, 73 G/s) and its two echoes (140 M, 33 M).  Ordering the work more finely
  // moves edges from the efficient sweep into rounds of the first kind and adds a pass over the bucket per cut; it removes relaxations,
  // not time.  What did help the wide rounds is more edges in flight per lane (expand_*_mlp: 11.3 -> 10.4 ms).  The single near queue
  // stays the default.
  char const* env_mode = getenv("CUGRAPH_AMD_SSSP_MODE");
  bool const use_dev   = !use_lh && env_mode && std::string(env_mode) == "dev";    // uniform sub-queues + device-driven rounds (k_sssp_ctl)
  bool const use_radix = !use_lh && env_mode && std::string(env_mode) == "radix";  // radix-heap sub-queues (see below)
  bool const use_multi = !use_lh && env_mode && (std::string(env_mode) == "multi" || use_dev || use_radix);
  static bool const sssp_trace_multi = getenv("CUGRAPH_AMD_SSSP_TRACE") != nullptr;
  uint64_t steps = 0, relaxed = 0;
  counters_t c;
  // fp32 + predecessors: (distance, parent) packed into one 64-bit word per vertex, lowered by one atomicMin per successful relaxation
  // (sssp_relax<WT, true>): no sweep over the settled edges afterwards.  Default schedule only; CUGRAPH_AMD_SSSP_PACKED=0 keeps the sweep.
  bool packed = false;
  dvec<unsigned long long> pk;
  if (use_multi) {
    std::vector<dvec<int32_t>> subq(SSSP_K);
    for (auto& q : subq) q.resize_discard(n1);
    dvec<bits_t> done(n1);
    hipLaunchKernelGGL(k_fill_t<bits_t>, grid_for(nv, kBlock, 4096), kBlock, 0, h.stream, done.data(), nv, unreached_bits);
    {
      bits_t zero_bits = 0;
      HIP_TRY(hipMemcpyAsync(d + source, &zero_bits, sizeof(bits_t), hipMemcpyHostToDevice, h.stream));
      HIP_TRY(hipMemcpyAsync(subq[0].data(), &source, 4, hipMemcpyHostToDevice, h.stream));
      h.sync();
    }
    char const* env_k = getenv("CUGRAPH_AMD_SSSP_SUBQ");  // uniform modes: sub-queues actually used (1 .. SSSP_K; 1 = the single near queue through these kernels)
    int const kk = use_radix ? SSSP_K : (env_k ? std::max(1, std::min(SSSP_K, atoi(env_k))) : 8);
    // the "work" of a sub-queue = its entries with at least average out-degree: ids are degree-sorted (renumbered graphs), so that is
    // "ids below heavy_cut"; counted once per graph
    if (g.sssp_heavy_cut < 0) {
      int64_t cut = nv;
      if (g.renumbered && nv > 0 && use_radix) {
        dvec<unsigned long long> n_ge(1);
        HIP_TRY(hipMemsetAsync(n_ge.data(), 0, 8, h.stream));
        hipLaunchKernelGGL(k_count_degree_at_least, grid_for(nv, kBlock, 2048), kBlock, 0, h.stream, row_beg, nv, (uint32_t)std::max(1.0, std::ceil(avg_deg)), n_ge.data());
        unsigned long long r = 0;
        h.read_back(&r, (unsigned long long const*)n_ge.data(), 1);
        cut = (int64_t)r;
      }
      g.sssp_heavy_cut = cut;
    }
    int32_t const heavy_cut = (int32_t)std::min<int64_t>(g.sssp_heavy_cut, INT32_MAX);
    double lower = 0.0;
    double ubh[SSSP_K];
    uint32_t tagh[SSSP_K], next_tag = 0x80000000u;
    // bucket 0 of the radix layout is delta / radix_div wide (delta = the reference's near / far width); the layout covers 2^(K-1) of those
    char const* env_div = getenv("CUGRAPH_AMD_SSSP_RADIX_DIV");
    double const delta0 = delta / (env_div ? std::max(1.0, atof(env_div)) : 256.0);
    char const* env_sm = getenv("CUGRAPH_AMD_SSSP_SPLIT_MIN");
    uint32_t const split_min = env_sm ? (uint32_t)std::max(1, atoi(env_sm)) : 2048u;
    auto set_uniform = [&](double lo, double hi) {
      lower = lo;
      for (int k = 0; k < SSSP_K; ++k) { ubh[k] = k < kk - 1 ? lo + (hi - lo) * (double)(k + 1) / (double)kk : hi; tagh[k] = next_tag++; }
    };
    auto set_radix = [&](double lo) {  // widths d0, d0, 2 d0, 4 d0, ...
      lower = lo;
      for (int k = 0; k < SSSP_K; ++k) { ubh[k] = lo + delta0 * std::ldexp(1.0, k); tagh[k] = next_tag++; }
    };
    if (use_radix) set_radix(0.0); else set_uniform(0.0, delta);
    uint32_t qn_h[SSSP_K] = {1u}, qh_h[SSSP_K] = {1u};
    int32_t* same_nxt = qa.data();
    int32_t* same_oth = qb.data();
    int32_t* far_cur  = fa.data();
    int32_t* far_nxt  = fb.data();
    int32_t const* front = nullptr;
    int64_t n_front = 0, n_far = 0;
    uint32_t round = 0, far_epoch = 1;
    int j = 0;
    auto state = [&](int jj) {
      sssp_multi_state<WT> s;
      s.dist = d; s.weights = w; s.done = done.data();
      for (int k = 0; k < SSSP_K; ++k) { s.q[k] = subq[k].data(); s.ub[k] = (WT)std::min(ubh[k], (double)wmax); s.tag[k] = tagh[k]; }
      s.q_same = same_nxt; s.far = far_cur; s.mark = mark_near.data(); s.mark_far = mark_far.data(); s.cnt = cnt.data();
      s.qn = &cnt.data()->padl0[0];  // the sub-queue fills live in the padding behind n_next: one read-back per round brings everything
      s.qh = &cnt.data()->padl1[0];  // ... their heavy counts behind n_far
      s.lower = (WT)std::min(lower, (double)wmax); s.cutoff = cutoff; s.heavy_cut = heavy_cut;
      s.j = jj; s.round_tag = round; s.far_epoch = far_epoch;
      return s;
    };
    auto upload_counters = [&]() {
      counters_t z{};
      z.n_far = (uint32_t)n_far;
      for (int k = 0; k < SSSP_K; ++k) { z.padl0[k] = qn_h[k]; z.padl1[k] = qh_h[k]; }
      z.far_min_bits_lo = 0xFFFFFFFFu;
      z.far_min_bits64  = ~0ull;
      std::memcpy(h.pinned, &z, sizeof(z));
      HIP_TRY(hipMemcpyAsync(cnt.data(), h.pinned, sizeof(z), hipMemcpyHostToDevice, h.stream));
    };
    auto download_counters = [&]() {
      h.read_back(&c, cnt.data(), 1);
      for (int k = 0; k < SSSP_K; ++k) {
        qn_h[k] = c.padl0[k]; qh_h[k] = c.padl1[k];
        CGA_EXPECTS((int64_t)qn_h[k] <= nv, CUGRAPH_UNKNOWN_ERROR, "sssp: sub-queue overflow");
      }
      c.fold();
      n_far = c.n_far;
      CGA_EXPECTS(n_far <= nv && (int64_t)c.n_next <= nv, CUGRAPH_UNKNOWN_ERROR, "sssp: queue overflow");
    };
    double far_min = 0.0;  // smallest distance the last split kept in the far pile
    auto read_far_min = [&]() {
      if constexpr (sizeof(WT) == 4) { float f; uint32_t b = c.far_min_bits_lo; std::memcpy(&f, &b, 4); far_min = f; }
      else { double f; unsigned long long b = c.far_min_bits64; std::memcpy(&f, &b, 8); far_min = f; }
    };
    // entries of `in` -> the sub-queues of the current bounds (distances below `lower`: stale, dropped; at or beyond the last bound: far pile)
    auto split_into_subqueues = [&](int32_t const* in, int64_t n_in, int32_t* far_out) {
      ++round;
      upload_counters();
      sssp_multi_state<WT> s = state(0);
      hipLaunchKernelGGL(k_sssp_split_multi<WT>, grid_for(n_in, TV_BLOCK, 2048), TV_BLOCK, 0, h.stream, in, n_in, s, far_out, far_epoch);
      download_counters();
      read_far_min();
    };
    // the sub-queues are exhausted: open the next window on the far pile
    auto advance_window = [&]() {
      double const upper = ubh[SSSP_K - 1];
      if (use_radix) set_radix(upper); else set_uniform(upper, upper + delta);
      for (;;) {
        ++far_epoch;
        int64_t const n_in = n_far;
        n_far = 0;
        split_into_subqueues(far_cur, n_in, far_nxt);
        std::swap(far_cur, far_nxt);
        bool any = false;
        for (int k = 0; k < SSSP_K; ++k) any |= qn_h[k] != 0;
        if (any || n_far == 0) break;
        // empty window: jump to the one holding the smallest far distance (never past it: fl(dmin / step) may round up to an integer)
        double const step = use_radix ? delta0 : delta;
        double kq = std::floor(far_min / step);
        while (kq > 0.0 && kq * step > far_min) kq -= 1.0;
        double const lo = std::max(kq * step, ubh[SSSP_K - 1]);
        if (use_radix) set_radix(lo); else set_uniform(lo, lo + delta);
      }
      j = 0;
    };
    auto one_round = [&](int jj) {
      ++round;
      ++steps;
      upload_counters();
      sssp_multi_state<WT> s = state(jj);
      {
        timed_launch t(h, "sssp_relax");
        hipLaunchKernelGGL(k_sssp_expand_multi<WT>, expand_grid(h, n_front), TV_BLOCK, 0, h.stream, front, n_front, row_beg, adj, bigq.data(), s, big_deg_for(h, n_front));
        hipLaunchKernelGGL(k_sssp_expand_big_multi<WT>, h.num_cus * 8, TV_BLOCK, 0, h.stream, (int32_t const*)bigq.data(), row_beg, adj, s);
      }
      download_counters();
      if (sssp_trace_multi) {
        static auto t_prev = std::chrono::steady_clock::now();
        auto const now = std::chrono::steady_clock::now();
        fprintf(stderr, "[sssp multi] round %3u  sub-queue %2d [%g, %g)  frontier %9lld  edges %11llu  same %9u  far %9u  %8.1f us\n", round, jj,
                jj ? ubh[jj - 1] : lower, ubh[jj], (long long)n_front, (unsigned long long)c.edges, c.n_next, c.n_far,
                std::chrono::duration<double, std::micro>(now - t_prev).count());
        t_prev = now;
      }
      relaxed += c.edges;
      front   = same_nxt;
      n_front = c.n_next;
      std::swap(same_nxt, same_oth);
    };
    if (use_radix) {
      // Radix-heap buckets (monotone: distances only ever enter sub-queues at or above the one being drained).  The first non-empty
      // sub-queue k is either drained as it is -- a frontier Bellman-Ford inside its range, cheap when it holds little work -- or, when
      // it holds at least split_min heavy vertices and is wider than delta0, cut into the k empty sub-queues below it (widths w / 2^(k-1),
      // w / 2^(k-1), w / 2^(k-2), ..., w / 2) by one pass over ITS entries only: the dense part of the distance range ends up in
      // sub-queues one delta0 wide (a vertex is expanded once, with its final distance), the sparse tail in a handful of wide ones.
      for (;;) {
        int k = 0;
        while (k < SSSP_K && qn_h[k] == 0) ++k;
        if (k == SSSP_K) {
          if (n_far == 0) break;
          advance_window();
          continue;
        }
        double const lb = k ? ubh[k - 1] : lower, wd = ubh[k] - lb;
        if (k > 0 && qh_h[k] >= split_min && wd > delta0 * 1.5) {
          int m = 1 + (int)std::floor(std::log2(wd / delta0) + 1e-6);  // finest sub-queue no narrower than delta0
          m     = std::max(2, std::min(m, k));
          lower = lb;
          for (int i = 0; i < k; ++i) {
            ubh[i]  = i < m - 1 ? lb + wd * std::ldexp(1.0, i - (m - 1)) : ubh[k];
            tagh[i] = next_tag++;
          }
          int64_t const n_in = qn_h[k];
          qn_h[k] = 0; qh_h[k] = 0;
          tagh[k] = next_tag++;  // (its range is empty now)
          split_into_subqueues(subq[k].data(), n_in, far_cur);
          if (sssp_trace_multi)
            fprintf(stderr, "[sssp multi] sub-queue %d [%g, %g): %lld entries cut into %d sub-queues\n", k, lb, ubh[k], (long long)n_in, m);
          continue;
        }
        front   = subq[k].data();
        n_front = qn_h[k];
        qn_h[k] = 0; qh_h[k] = 0;
        while (n_front > 0) one_round(k);
        lower = ubh[k];
      }
    } else if (use_dev) {
      char const* env_b = getenv("CUGRAPH_AMD_SSSP_BATCH");  // rounds enqueued per host synchronisation
      int const batch   = env_b ? std::max(1, atoi(env_b)) : 8;
      int const grid    = h.num_cus * 8;
      for (;;) {
        int j0 = 0;
        while (j0 < SSSP_K && qn_h[j0] == 0) ++j0;
        if (j0 == SSSP_K) {
          if (n_far == 0) break;
          advance_window();
          continue;
        }
        ++round;
        sssp_dev_args<WT> a;
        a.s            = state(j0);
        a.qsame[0]     = qa.data();
        a.qsame[1]     = qb.data();
        a.narrow_limit = (int32_t)std::min<int64_t>((int64_t)h.num_cus * 64, INT32_MAX);
        a.kk           = kk;
        {
          counters_t z{};
          z.n_far = (uint32_t)n_far;
          for (int k = 0; k < SSSP_K; ++k) z.padl0[k] = k == j0 ? 0u : qn_h[k];
          z.far_min_bits_lo = 0xFFFFFFFFu;
          z.far_min_bits64  = ~0ull;
          sssp_ctl_t c0{};
          c0.j = (uint32_t)j0; c0.n_front = qn_h[j0]; c0.round = round;
          *sssp_ctl_of(&z) = c0;
          std::memcpy(h.pinned, &z, sizeof(z));
          HIP_TRY(hipMemcpyAsync(cnt.data(), h.pinned, sizeof(z), hipMemcpyHostToDevice, h.stream));
        }
        sssp_ctl_t ctl{};
        for (;;) {
          {
            timed_launch t(h, "sssp_relax");
            for (int b = 0; b < batch; ++b) {
              hipLaunchKernelGGL(k_sssp_expand_dev<WT>, grid, TV_BLOCK, 0, h.stream, row_beg, adj, bigq.data(), a);
              hipLaunchKernelGGL(k_sssp_expand_big_dev<WT>, grid, TV_BLOCK, 0, h.stream, (int32_t const*)bigq.data(), row_beg, adj, a);
              hipLaunchKernelGGL(k_sssp_ctl, 1, 64, 0, h.stream, cnt.data(), SSSP_K);
            }
          }
          h.read_back(&c, cnt.data(), 1);
          ctl = *sssp_ctl_of(&c);
          CGA_EXPECTS((int64_t)c.n_far <= nv && (int64_t)c.n_next <= nv, CUGRAPH_UNKNOWN_ERROR, "sssp: queue overflow");
          for (int k = 0; k < SSSP_K; ++k) CGA_EXPECTS((int64_t)c.padl0[k] <= nv, CUGRAPH_UNKNOWN_ERROR, "sssp: sub-queue overflow");
          if (ctl.done) break;
        }
        steps += ctl.rounds;
        relaxed += ctl.relaxed;
        round = ctl.round;
        n_far = c.n_far;
        for (int k = 0; k < SSSP_K; ++k) { qn_h[k] = 0; qh_h[k] = 0; }
      }
    } else {
      for (;;) {
        if (n_front == 0) {
          while (j < SSSP_K && qn_h[j] == 0) ++j;
          if (j < SSSP_K) {  // drain the next non-empty sub-queue
            front   = subq[j].data();
            n_front = qn_h[j];
            qn_h[j] = 0; qh_h[j] = 0;
          } else {
            if (n_far == 0) break;
            advance_window();
            continue;
          }
        }
        one_round(j);
      }
    }
  } else {
  {  // d[source] = 0, near = {source}
    bits_t zero_bits = 0;
    HIP_TRY(hipMemcpyAsync(d + source, &zero_bits, sizeof(bits_t), hipMemcpyHostToDevice, h.stream));
    HIP_TRY(hipMemcpyAsync(qa.data(), &source, 4, hipMemcpyHostToDevice, h.stream));
    if (use_lh) {  // the source is the first member of the first bucket
      uint32_t const one = 1;
      HIP_TRY(hipMemcpyAsync(sa.data(), &source, 4, hipMemcpyHostToDevice, h.stream));
      HIP_TRY(hipMemcpyAsync(mark_set.data() + source, &one, 4, hipMemcpyHostToDevice, h.stream));
    }
    h.sync();
  }
  int32_t* q_cur = qa.data();
  int32_t* q_nxt = qb.data();
  int32_t* far_cur = fa.data();
  int32_t* far_nxt = fb.data();
  int64_t n_cur = 1, n_far = 0, n_set = use_lh ? 1 : 0;
  int32_t* set_cur = sa.data();
  int32_t* set_nxt = sb.data();
  uint32_t round = 0, far_epoch = 1, set_epoch = 1;
  double lower = 0.0, upper = delta;
  // one relaxation round over the rows [beg[u], end[u]) of the vertices in `front`; near / far / bucket-member appends continue
  // at n_far / n_set_in (they persist across the rounds of a bucket), n_next / n_big / edges start from zero
  // pull rounds (sssp_pull_fn): chosen from the frontier's out-edge count, which the round that built the frontier summed on the device
  // OPT-IN (CUGRAPH_AMD_SSSP_PULL=1; "force": every round, the parity test): measured at RMAT-24, weights 1..255 (profiles/r4o_sssp_pull.txt): the
  // round of the 66 K hubs right after the source takes 2.7 ms pulled (268 M in-edges streamed, 82 M of them relaxed; the rows of 2048 or
  // more in-edges -- 82 M edges -- still go through the atomic path) against 2.5 ms pushed, and a traversal 12.1 ms against 11.0: no gain.
  char const* env_pull     = getenv("CUGRAPH_AMD_SSSP_PULL");
  bool const pull_allowed  = !use_lh && g.ne <= kMaxSignedEdges && env_pull && std::string(env_pull) != "0";
  bool const pull_force    = pull_allowed && env_pull && std::string(env_pull) == "force";
  uint64_t const pull_min_edges = std::max<uint64_t>((uint64_t)g.ne / 10, (uint64_t)1 << 22);
  uint64_t front_edges = 0;  // out-edges of the current near frontier (0 for the source: its round is a push)
  dvec<uint32_t> fbits;
  {
    char const* env_pk = getenv("CUGRAPH_AMD_SSSP_PACKED");
    packed = compute_predecessors && sizeof(WT) == 4 && !use_lh && !pull_allowed && !(env_pk && std::string(env_pk) == "0");
    if (packed) {
      pk.resize_discard(n1);
      unsigned long long const none = ((unsigned long long)(uint32_t)unreached_bits << 32) | 0x7FFFFFFFull;
      hipLaunchKernelGGL(k_fill_t<unsigned long long>, grid_for(nv, kBlock, 4096), kBlock, 0, h.stream, pk.data(), nv, none);
      unsigned long long const at_source = 0x7FFFFFFFull;  // distance 0, no parent
      HIP_TRY(hipMemcpyAsync(pk.data() + source, &at_source, 8, hipMemcpyHostToDevice, h.stream));
      h.sync();  // (at_source is a local)
    }
  }
  static bool const sssp_trace = getenv("CUGRAPH_AMD_SSSP_TRACE") != nullptr;  // per round: sizes and wall time since the previous line (stderr)
  auto t_trace = std::chrono::steady_clock::now();
  // The hub round -- the few ten thousand hubs right after the source, whose out-edges are a third of the graph and whose relaxations mostly
  // SUCCEED (a vertex reached from k hubs is lowered ~ln k times when they arrive in any order) -- takes its frontier ordered by tentative
  // distance (256 bands of the window, one 8-bit radix pass): the closest hubs go first, later candidates mostly fail the cheap pre-test.
  // RMAT-24, weights 1..255, 16 roots, same session: 12.19 -> 11.78 ms with predecessors, 11.04 -> 10.74 without (profiles/r5g_sssp_sort.txt);
  // ordering EVERY wide round costs more (the sorts) than the 5 % of relaxations it saves (r5f_sssp_sort.txt).  CUGRAPH_AMD_SSSP_SORT=0: off,
  // =n: every round of at least n vertices (the experiment).
  char const* env_sort = getenv("CUGRAPH_AMD_SSSP_SORT");
  bool const sort_hub_only = env_sort == nullptr || std::string(env_sort) == "hub";
  int64_t const sort_min = sort_hub_only ? 1024 : std::max<int64_t>(0, atoll(env_sort));  // 0 = off
  dvec<uint64_t> sk, sk_out;
  dvec<uint32_t> sv, sv_out, sort_hist;
  auto relax_round = [&](int32_t const* front, int64_t n_front, int32_t const* beg, int32_t const* end, int32_t* set_out, int64_t n_set_in) {
    ++round;
    ++steps;
    bool const hub_round = front_edges >= std::max<uint64_t>((uint64_t)g.ne / 10, (uint64_t)1 << 22) && n_front * 16 <= nv;  // few vertices, a large share of the edges
    if (sort_min > 0 && n_front >= sort_min && !use_lh && (sort_hub_only ? hub_round && !g.weights_uniform : true)) {
      if ((int64_t)sk.size() < n_front) {
        sk.resize_discard((size_t)n_front); sk_out.resize_discard((size_t)n_front); sv.resize_discard((size_t)n_front); sv_out.resize_discard((size_t)n_front);
        sort_hist.resize_discard(radix_pass_scratch(n_front));
      }
      WT const lo = (WT)std::min(lower, (double)wmax), inv = (WT)(256.0 / delta);
      bool done = false;
      if constexpr (sizeof(WT) == 4) {
        if (packed) {
          hipLaunchKernelGGL((k_sssp_band_keys<WT, true>), grid_for(n_front, kBlock, 4096), kBlock, 0, h.stream, front, n_front, reinterpret_cast<bits_t const*>(pk.data()), lo, inv, sk.data(), sv.data());
          done = true;
        }
      }
      if (!done) hipLaunchKernelGGL((k_sssp_band_keys<WT, false>), grid_for(n_front, kBlock, 4096), kBlock, 0, h.stream, front, n_front, (bits_t const*)d, lo, inv, sk.data(), sv.data());
      radix_pass_u64_u32(h, sk.data(), sv.data(), sk_out.data(), sv_out.data(), n_front, 0, 8, sort_hist.data());
      front = reinterpret_cast<int32_t const*>(sv_out.data());
    }
    counters_t z{};
    z.n_far           = (uint32_t)n_far;
    z.n_set           = (uint32_t)n_set_in;
    z.far_min_bits_lo = 0xFFFFFFFFu;
    z.far_min_bits64  = ~0ull;
    std::memcpy(h.pinned, &z, sizeof(z));
    HIP_TRY(hipMemcpyAsync(cnt.data(), h.pinned, sizeof(z), hipMemcpyHostToDevice, h.stream));
    sssp_state<WT> s{d, wrel, q_nxt, far_cur, mark_near.data(), mark_far.data(), use_lh ? set_out : nullptr, mark_set.data(), set_epoch, cnt.data(),
                     (WT)std::min(upper, (double)wmax), cutoff, round, far_epoch, (int32_t const*)o.offsets.data()};
    if (packed) { s.pk = pk.data(); s.labels = g.renumbered ? g.number_map.data() : nullptr; s.source = source; }
    // few vertices with a large share of the graph's edges (the hubs right after the source): a pull round over the in-edges (sssp_pull_fn)
    bool const pull = pull_allowed && (pull_force || (front_edges >= pull_min_edges && n_front * 16 <= nv));
    if (pull) {
      ensure_orientation(h, g, true);  // in-edges + their weights: built once per graph, next to the CSR under the same numbering
      orientation_t const& ci = g.csc;
      size_t const fw = (size_t)((nv + 31) / 32 + 1);
      if (fbits.size() < fw) fbits.resize_discard(fw);
      HIP_TRY(hipMemsetAsync(fbits.data(), 0, fw * 4, h.stream));
      hipLaunchKernelGGL(k_queue_to_bits, grid_for(n_front, kBlock, 2048), kBlock, 0, h.stream, front, n_front, fbits.data());
      s.weights = ci.weights.template as<WT const>();
      {
        timed_launch t(h, "sssp_relax");
        int64_t const ngroup = (nv + 63) / 64;
        int const grid       = (int)std::max<int64_t>(1, std::min<int64_t>((ngroup + TV_WAVES - 1) / TV_WAVES, (int64_t)h.num_cus * 16));
        hipLaunchKernelGGL(k_sssp_pull_rows<WT>, grid, TV_BLOCK, 0, h.stream, (int32_t const*)ci.offsets.data(), (int32_t const*)ci.indices.data(), ci.weights.template as<WT const>(), nv,
                           (uint32_t const*)fbits.data(), bigq.data(), s, (int32_t)BIG_DEG);
        hipLaunchKernelGGL(k_sssp_pull_big<WT>, h.num_cus * 8, TV_BLOCK, 0, h.stream, (int32_t const*)bigq.data(), (int32_t const*)ci.offsets.data(), (int32_t const*)ci.indices.data(), s,
                           (uint32_t const*)fbits.data());
      }
    } else {
      timed_launch t(h, "sssp_relax");
      bool launched = false;
      if constexpr (sizeof(WT) == 4) {
        if (packed) {
          hipLaunchKernelGGL((k_sssp_expand<WT, true>), expand_grid(h, n_front), TV_BLOCK, 0, h.stream, front, n_front, beg, end, adj, bigq.data(), s, big_deg_for(h, n_front));
          hipLaunchKernelGGL((k_sssp_expand_big<WT, true>), h.num_cus * 8, TV_BLOCK, 0, h.stream, (int32_t const*)bigq.data(), beg, end, adj, s);
          launched = true;
        }
      }
      if (!launched) {
        hipLaunchKernelGGL(k_sssp_expand<WT>, expand_grid(h, n_front), TV_BLOCK, 0, h.stream, front, n_front, beg, end, adj, bigq.data(), s, big_deg_for(h, n_front));
        hipLaunchKernelGGL(k_sssp_expand_big<WT>, h.num_cus * 8, TV_BLOCK, 0, h.stream, (int32_t const*)bigq.data(), beg, end, adj, s);
      }
    }
    h.read_back(&c, cnt.data(), 1);
    c.fold();
    if (pull) c.edges = front_edges;  // relaxations = the frontier's out-edges (the kernel walked every in-edge of the graph to find them)
    front_edges = c.out_edges;
    if (sssp_trace) {
      auto const now = std::chrono::steady_clock::now();
      fprintf(stderr, "[sssp] round %3u %s window [%g, %g)  frontier %9lld  edges %11llu  next %9u (%llu out-edges)  far %9u  deferred segments %7u  %8.1f us\n", round,
              pull ? "PULL" : "push", lower, upper, (long long)n_front, (unsigned long long)c.edges, c.n_next, (unsigned long long)c.out_edges, c.n_far, c.n_big,
              std::chrono::duration<double, std::micro>(now - t_trace).count());
      t_trace = now;
    }
    relaxed += c.edges;
    n_cur = c.n_next;
    n_far = c.n_far;
    CGA_EXPECTS(n_far <= nv, CUGRAPH_UNKNOWN_ERROR, "sssp: far pile overflow");
    std::swap(q_cur, q_nxt);
  };
  for (;;) {
    for (;;) {  // one bucket
      while (n_cur > 0) {
        relax_round(q_cur, n_cur, row_beg, lend, set_cur, n_set);  // lend == nullptr: all edges (wide buckets)
        if (use_lh) n_set = c.n_set;
      }
      if (!use_lh || n_set == 0) break;
      // the bucket is closed: its members' distances are final -- their heavy edges, once.  (A heavy edge cannot land inside the
      // bucket: fl(d + w) >= fl(lower + delta) = upper; should it ever, the vertex goes to the next set and the loop runs again.)
      ++set_epoch;
      int64_t const n_members = n_set;
      relax_round(set_cur, n_members, lend, row_beg + 1, set_nxt, 0);
      n_set = c.n_set;
      std::swap(set_cur, set_nxt);
      if (n_cur == 0 && n_set == 0) break;
    }
    if (n_far == 0) break;
    // advance the bucket window until the far pile yields a non-empty near frontier
    while (n_cur == 0 && n_far > 0) {
      lower = upper;
      upper = upper + delta;
      ++round;
      ++far_epoch;
      counters_t z{};
      z.far_min_bits_lo = 0xFFFFFFFFu;
      z.far_min_bits64  = ~0ull;
      std::memcpy(h.pinned, &z, sizeof(z));
      HIP_TRY(hipMemcpyAsync(cnt.data(), h.pinned, sizeof(z), hipMemcpyHostToDevice, h.stream));
      bool split_done = false;
      if constexpr (sizeof(WT) == 4) {
        if (packed) {
          hipLaunchKernelGGL((k_sssp_split<WT, true>), grid_for(n_far, TV_BLOCK, 2048), TV_BLOCK, 0, h.stream, (int32_t const*)far_cur, n_far,
                             reinterpret_cast<bits_t const*>(pk.data()), (WT)std::min(lower, (double)wmax), (WT)std::min(upper, (double)wmax), q_cur, far_nxt,
                             mark_near.data(), mark_far.data(), round, far_epoch, cnt.data(), (int32_t*)nullptr, mark_set.data(), set_epoch,
                             (int32_t const*)o.offsets.data());
          split_done = true;
        }
      }
      if (!split_done)
      hipLaunchKernelGGL(k_sssp_split<WT>, grid_for(n_far, TV_BLOCK, 2048), TV_BLOCK, 0, h.stream, (int32_t const*)far_cur, n_far,
                         (bits_t const*)d, (WT)std::min(lower, (double)wmax), (WT)std::min(upper, (double)wmax), q_cur, far_nxt,
                         mark_near.data(), mark_far.data(), round, far_epoch, cnt.data(), use_lh ? set_cur : (int32_t*)nullptr, mark_set.data(), set_epoch,
                         (int32_t const*)o.offsets.data());
      h.read_back(&c, cnt.data(), 1);
      c.fold();
      front_edges = c.out_edges;
      n_cur = c.n_next;
      n_far = c.n_far;
      n_set = use_lh ? c.n_set : 0;
      std::swap(far_cur, far_nxt);
      if (n_cur == 0 && n_far > 0) {  // empty buckets: jump to the one holding the smallest far distance
        double dmin;
        if constexpr (sizeof(WT) == 4) { float f; uint32_t b = c.far_min_bits_lo; std::memcpy(&f, &b, 4); dmin = f; }
        else { double f; unsigned long long b = c.far_min_bits64; std::memcpy(&f, &b, 8); dmin = f; }
        double k = std::floor(dmin / delta);
        while (k > 0.0 && k * delta > dmin) k -= 1.0;  // fl(dmin / delta) may round up to an integer: never jump past the smallest far distance
        if (k * delta > upper) upper = k * delta;
      }
    }
  }

  }

  if (packed) {  // the two result columns out of the packed words; nothing else to do for the parents
    if constexpr (sizeof(WT) == 4) {
      if (nv > 0) hipLaunchKernelGGL(k_sssp_unpack, grid_for(nv, kBlock, 4096), kBlock, 0, h.stream, (unsigned long long const*)pk.data(), nv, reinterpret_cast<uint32_t*>(d), preds->buf.as<int32_t>());
    }
    c = counters_t{};
  } else if (compute_predecessors) {
    fill_i32(h, preds->buf.as<int32_t>(), nv, INT32_MAX);
    HIP_TRY(hipMemsetAsync(cnt.data(), 0, sizeof(counters_t), h.stream));
    sssp_parent<WT> f{(bits_t const*)d, w, preds->buf.as<int32_t>(), g.renumbered ? g.number_map.data() : nullptr, source};
    keep_reached<WT> keep{(bits_t const*)d, unreached_bits};
    if (nv > 0) {
      hipLaunchKernelGGL(k_sssp_parents<WT>, expand_grid(h, nv), TV_BLOCK, 0, h.stream, nv, (int32_t const*)o.offsets.data(),
                         (int32_t const*)o.indices.data(), bigq.data(), cnt.data(), f, keep);
      hipLaunchKernelGGL(k_sssp_parents_big<WT>, h.num_cus * 8, TV_BLOCK, 0, h.stream, (int32_t const*)bigq.data(), (int32_t const*)o.offsets.data(),
                         (int32_t const*)o.indices.data(), cnt.data(), f);
    }
    if (nv > 0) hipLaunchKernelGGL(k_fix_pred, grid_for(nv, kBlock, 4096), kBlock, 0, h.stream, preds->buf.as<int32_t>(), nv);
    h.read_back(&c, cnt.data(), 1);
    c.fold();
  }
  dvec<unsigned long long> reached(1);
  HIP_TRY(hipMemsetAsync(reached.data(), 0, 8, h.stream));
  if (nv > 0) hipLaunchKernelGGL(k_count_reached_dist<bits_t>, grid_for(nv, kBlock, 1024), kBlock, 0, h.stream, (bits_t const*)d, nv, unreached_bits, reached.data());
  unsigned long long nreached;
  h.read_back(&nreached, reached.data(), 1);
  h.last_stats = cugraph_amd_traversal_stats_t{steps, relaxed, nreached, compute_predecessors ? c.edges : 0};
  if (nv > 0) HIP_TRY(hipMemcpyAsync(ids->buf.ptr, g.number_map.data(), nv * 4, hipMemcpyDeviceToDevice, h.stream));
  h.sync();
  auto* r = new paths_result_t{ids.release(), dist.release(), preds.release()};
  outer_replace_ids(h, g, r->vertex_ids);
  outer_replace_ids(h, g, r->predecessors);
  return r;
}

}  // namespace
}  // namespace cga

using namespace cga;

extern "C" cugraph_error_code_t cugraph_bfs(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
                                            cugraph_type_erased_device_array_view_t* sources, bool_t direction_optimizing, size_t depth_limit,
                                            bool_t compute_predecessors, bool_t /*do_expensive_check*/, cugraph_paths_result_t** result,
                                            cugraph_error_t** error)
{
  if (result) *result = nullptr;
  return guarded(error, [&] {
    CGA_EXPECTS(result != nullptr, CUGRAPH_INVALID_INPUT, "result is NULL");
    handle_t& h = const_cast<handle_t&>(H(handle));
    graph_t& g  = GM(graph);
    if (g.mg) {  // a graph from cugraph_graph_create_mg on a communicator handle: collective (traversal_mg_driver.hip)
      vertex_column_in c_msources;  // INT64 ids of a multi-GPU graph: compact int32 ids in, the caller's ids out (outer_ids.hip)
      device_array_view_t const* msv = c_msources.get(h, g, V(sources), "sources");
      paths_result_t* r = mg_run_bfs(h, g, msv, direction_optimizing == TRUE, depth_limit, compute_predecessors == TRUE);
      outer_replace_ids(h, g, r->vertex_ids);
      outer_replace_dist(h, g, r->distances);  // BFS distances carry the vertex type (bfs.cpp:156-187)
      outer_replace_ids(h, g, r->predecessors);
      *result = reinterpret_cast<cugraph_paths_result_t*>(r);
      return;
    }
    *result     = reinterpret_cast<cugraph_paths_result_t*>(
      run_bfs(h, g, V(sources), direction_optimizing == TRUE, depth_limit, compute_predecessors == TRUE));
  });
}

extern "C" cugraph_error_code_t cugraph_sssp(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph, size_t source, double cutoff,
                                             bool_t compute_predecessors, bool_t /*do_expensive_check*/, cugraph_paths_result_t** result,
                                             cugraph_error_t** error)
{
  if (result) *result = nullptr;
  return guarded(error, [&] {
    CGA_EXPECTS(result != nullptr, CUGRAPH_INVALID_INPUT, "result is NULL");
    handle_t& h = const_cast<handle_t&>(H(handle));
    graph_t& g  = GM(graph);
    if (g.mg) {
      size_t msource = source;
      if (g.outer.active) {  // INT64 ids: the source's position in the sorted id list every rank holds (no collective: all ranks get the same answer)
        dvec<int64_t> one(1);
        dvec<int32_t> compact(1);
        int64_t const sv = (int64_t)source;
        HIP_TRY(hipSetDevice(h.device));
        HIP_TRY(hipMemcpyAsync(one.data(), &sv, 8, hipMemcpyHostToDevice, h.stream));
        outer_to_compact(h, g.outer, one.data(), INT64, 1, compact.data());
        int32_t c = -1;
        h.read_back(&c, compact.data(), 1);
        CGA_EXPECTS(c >= 0, CUGRAPH_INVALID_INPUT, "Invalid input argument: source vertex is not a vertex of the graph.");
        msource = (size_t)c;
      }
      paths_result_t* r = mg_run_sssp(h, g, msource, cutoff, compute_predecessors == TRUE);
      outer_replace_ids(h, g, r->vertex_ids);
      outer_replace_ids(h, g, r->predecessors);
      *result = reinterpret_cast<cugraph_paths_result_t*>(r);
      return;
    }
    // sssp.cpp:72-73,105 dereferences the edge weights unconditionally: an unweighted graph is an error
    CGA_EXPECTS(g.has_weights, CUGRAPH_INVALID_INPUT, "cugraph_sssp requires a weighted graph");
    paths_result_t* r = g.weight_type == FLOAT64 ? run_sssp<double>(h, g, source, cutoff, compute_predecessors == TRUE)
                                                 : run_sssp<float>(h, g, source, cutoff, compute_predecessors == TRUE);
    *result = reinterpret_cast<cugraph_paths_result_t*>(r);
  });
}

extern "C" cugraph_type_erased_device_array_view_t* cugraph_paths_result_get_vertices(cugraph_paths_result_t* result)
{
  return reinterpret_cast<cugraph_type_erased_device_array_view_t*>(reinterpret_cast<paths_result_t*>(result)->vertex_ids->new_view());
}
extern "C" cugraph_type_erased_device_array_view_t* cugraph_paths_result_get_distances(cugraph_paths_result_t* result)
{
  return reinterpret_cast<cugraph_type_erased_device_array_view_t*>(reinterpret_cast<paths_result_t*>(result)->distances->new_view());
}
extern "C" cugraph_type_erased_device_array_view_t* cugraph_paths_result_get_predecessors(cugraph_paths_result_t* result)
{
  return reinterpret_cast<cugraph_type_erased_device_array_view_t*>(reinterpret_cast<paths_result_t*>(result)->predecessors->new_view());
}
extern "C" void cugraph_paths_result_free(cugraph_paths_result_t* result)
{
  auto r = reinterpret_cast<paths_result_t*>(result);
  if (!r) return;
  delete r->vertex_ids;
  delete r->distances;
  delete r->predecessors;
  delete r;
}
