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
  int32_t* pred;               // parent, -1 = none; the vertices a top-down level discovers hold the parent's INTERNAL id until the level's
                               // k_bfs_relabel_queue, everything else the caller's id already; nullptr when not requested
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
      // minimum internal id among the frontier parents; "none" is -1 = the largest unsigned word, so the array needs no other initial value
      uint32_t* const pw = reinterpret_cast<uint32_t*>(&s.pred[v]);
      if (s.pred && (uint32_t)u < __hip_atomic_load(pw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(pw, (uint32_t)u);
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

// The parents of the vertices a top-down level has just discovered (the level's output queue): internal ids -> the caller's ids.  Runs right after
// the level, on the queue while it is hot: the end of the traversal has no pass over all V parents any more (it took 110 us of a 1.3 ms search
// at RMAT-24).  Bottom-up levels write the caller's id at the discovery (k_bfs_bottom_up: parent_label).  n comes from the device counter.
__global__ void __launch_bounds__(256) k_bfs_relabel_queue(int32_t const* q, counters_t const* cnt, int32_t* pred, int32_t const* labels)
{
  int64_t const n = (int64_t)cnt->n_next;
  int64_t i       = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride  = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    int32_t const v = q[i], p = pred[v];
    if (p >= 0) pred[v] = labels[p];
  }
}

// the vertex column of the result: a copy of the numbering (the result owns its columns).  Runs on the handle's side stream beside the traversal.
__global__ void __launch_bounds__(256) k_copy_i32x4(int32_t const* in, int32_t* out, int64_t n)
{
  int64_t const t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  int64_t const n4 = n / 4;
  for (int64_t i = t; i < n4; i += stride) reinterpret_cast<int4*>(out)[i] = reinterpret_cast<int4 const*>(in)[i];
  for (int64_t i = n4 * 4 + t; i < n; i += stride) out[i] = in[i];
}

// dist / pred <- "unreached", the bitmaps and the counters <- 0
__global__ void __launch_bounds__(256) k_bfs_init_state(int32_t* dist, int32_t* pred, int64_t nv, uint32_t* vis_prev, uint32_t* vis_new, uint32_t* front, uint32_t* next,
                                                        int64_t nwords, counters_t* cnt)
{
  int64_t const t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  int64_t const n4 = nv / 4;
  int4 const big{INT32_MAX, INT32_MAX, INT32_MAX, INT32_MAX}, none{-1, -1, -1, -1};
  for (int64_t i = t; i < n4; i += stride) {
    reinterpret_cast<int4*>(dist)[i] = big;
    if (pred) reinterpret_cast<int4*>(pred)[i] = none;
  }
  for (int64_t i = n4 * 4 + t; i < nv; i += stride) { dist[i] = INT32_MAX; if (pred) pred[i] = -1; }
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

// The distance filter of a wide relaxation round (round 6).  What bounds such a round is not bytes but REQUESTS: every edge probes the
// tentative distance of its destination -- 4 or 8 bytes at a random place of a 64 / 128 MB array (RMAT-24) that the 4 MiB L2 of an XCD cannot
// hold, so each probe is a request to the fabric (66-85 G probes/s whatever the schedule, profiles/r5*), and 2.26 of them per edge fail.  The filter is one BIT per vertex,
// fbits[v] = (d[v] < T) for a threshold T chosen per round (sssp_filter_pick): 2 MB at RMAT-24, L2-resident.  A relaxation with candidate
// nd >= T into a vertex whose bit is set cannot succeed (d only decreases, so d[v] < T <= nd still holds; strictly, so the lexicographic
// (distance, parent) minimum cannot change either) and is dropped BEFORE the probe.  Exact for any T; T only decides how much is filtered.
template <typename WT>
struct sssp_filter {
  uint32_t const* bits{nullptr};  // nullptr: no filter in this round
  WT const* t{nullptr};           // the threshold the bits were built with (device-resident: chosen by a kernel, no host round trip)
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
  counters_t* cnt;
  WT threshold;        // near / far split
  WT cutoff;
  uint32_t round;
  uint32_t far_epoch;
  int32_t const* out_offsets;  // CSR offsets: the out-degrees of the vertices that enter the near frontier are summed (counters_t::out_edges):
                               // the host knows the next round's edge count without a pass over the queue
  // fp32 with predecessors (sssp_relax<WT, true>): (distance bits << 32 | external id of the parent) per vertex, lowered by ONE 64-bit
  // atomicMin -- the reference's reduction, a lexicographic minimum over (distance, predecessor) (sssp_impl.cuh:334), in the relaxation itself
  // instead of a sweep over the settled edges afterwards; `dist` is not used then
  unsigned long long* pk{nullptr};
  int32_t const* labels{nullptr};  // internal -> external id (nullptr: identity)
  int32_t source{-1};              // keeps its parent -1 whatever reaches it at distance 0
  sssp_filter<WT> flt{};
};
__device__ __forceinline__ uint32_t pk_dist_bits(unsigned long long const* pk, int32_t v) { return reinterpret_cast<uint32_t const*>(pk)[2 * (size_t)v + 1]; }
// parent labels inside the packed word are biased so that the UNSIGNED order of the word is the signed order of the external ids (a negative
// external id must not lose against every non-negative one: the sweep and the fp64 path take a signed minimum); k_sssp_unpack removes the bias
__device__ __forceinline__ uint32_t pk_label(int32_t label) { return (uint32_t)label ^ 0x80000000u; }
constexpr unsigned long long kPkNoParent = 0xFFFFFFFFull;  // low word of a vertex nobody has reached through an edge (the source keeps it)

template <typename WT, bool PK = false>
struct sssp_relax {
  static_assert(!PK || sizeof(WT) == 4, "the packed (distance, parent) word holds an fp32 distance");
  sssp_state<WT> s;
  wave_queue wq_near, wq_far;
  WT ft{0};                      // filter threshold (read once per workgroup; meaningless without s.flt.bits)
  unsigned long long deg_acc{0};
  unsigned long long probes{0};  // relaxations that went on to probe the destination's distance (reported as counters_t::in_edges)
  __device__ __forceinline__ void begin() { if (s.flt.bits) ft = *s.flt.t; }
  __device__ __forceinline__ void count_near(bool near, int32_t v)
  {
    if (near && s.out_offsets) deg_acc += (unsigned long long)(eoff(s.out_offsets, v + 1) - eoff(s.out_offsets, v));
  }
  __device__ __forceinline__ void flush()
  {
    wq_near.flush(); wq_far.flush();
    unsigned long long a = deg_acc, b = probes;
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if ((threadIdx.x & 63) == 0 && a) atomicAdd(&cnt_replica(s.cnt)->out_edges, a);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&cnt_replica(s.cnt)->in_edges, b);
  }
  __device__ __forceinline__ void operator()(int32_t u, int32_t v, eoff_t p)
  {  // the one-edge-at-a-time expansion: the phased form below, called in sequence
    cand_t const c = pre(u, v, p);
    tok_t const t  = mid(v, c);
    post(u, v, c, t, mid2(v, c, t));
  }
  // the phased form of the relaxation (expand_*_mlp: EX_U edges in flight per lane)
  struct cand_t { WT nd; bool pass; unsigned long long word; };  // word: the packed candidate (PK only)
  using tok_t = typename std::conditional<PK, unsigned long long, typename dist_bits<WT>::type>::type;
  // true: the relaxation cannot succeed, decided from the L2-resident bit (sssp_filter)
  __device__ __forceinline__ bool filtered(int32_t vv, WT nd) const
  {
    if (!s.flt.bits) return false;  // (wave-uniform)
    uint32_t const w = s.flt.bits[(uint32_t)vv >> 5];
    return ((w >> ((uint32_t)vv & 31u)) & 1u) != 0u && nd >= ft;
  }
  __device__ __forceinline__ cand_t pre(int32_t u, int32_t v, eoff_t p)
  {  // the loads of the EX_U edges of a step go out together.  d[v] is read with an agent-scope load: L2-served, so once a
     // hub has been lowered the other relaxations of this round see it and skip the atomic (a non-temporal load is L2-served too, but
     // its lines are not retained: 14.0 ms instead of 10.4 per SSSP at RMAT-24)
    using B = dist_bits<WT>;
    int32_t const uu = u < 0 ? 0 : u, vv = v < 0 ? 0 : v;
    if constexpr (PK) {
      WT const nd = B::from(pk_dist_bits(s.pk, uu)) + s.weights[p];
      unsigned long long const word = ((unsigned long long)B::to(nd) << 32) | pk_label(s.labels ? s.labels[uu] : uu);
      bool const go = (v >= 0) & (v != s.source) & (nd < s.cutoff) && !filtered(vv, nd);
      unsigned long long cur = 0ull;
      if (go) cur = __hip_atomic_load(&s.pk[vv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      probes += go ? 1ull : 0ull;
      // strictly smaller (distance, parent): a shorter distance, or the same distance through a parent with a smaller external id
      return cand_t{nd, go && word < cur, word};
    } else {
      WT const nd = B::from(s.dist[uu]) + s.weights[p];
      bool const go = (v >= 0) & (nd < s.cutoff) && !filtered(vv, nd);
      WT dv = WT(0);
      if (go) dv = B::from(__hip_atomic_load(&s.dist[vv], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      probes += go ? 1ull : 0ull;
      return cand_t{nd, go && nd < dv, 0ull};
    }
  }
  // the probe of a candidate that has passed the cheap tests already (sssp_two_stage::drain); v < 0: no candidate
  __device__ __forceinline__ cand_t probe(int32_t v, WT nd, uint32_t biased_label)
  {
    using B       = dist_bits<WT>;
    bool const go = v >= 0;
    probes += go ? 1ull : 0ull;
    if constexpr (PK) {
      unsigned long long const word = ((unsigned long long)B::to(nd) << 32) | biased_label;
      unsigned long long cur = 0ull;
      if (go) cur = __hip_atomic_load(&s.pk[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return cand_t{nd, go && word < cur, word};
    } else {
      WT dv = WT(0);
      if (go) dv = B::from(__hip_atomic_load(&s.dist[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      return cand_t{nd, go && nd < dv, 0ull};
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
    count_near(near, v);
    wq_near.push(near, v);
    wq_far.push(far, v);
  }
};

template <typename WT, bool PK = false>
__global__ void __launch_bounds__(TV_BLOCK) k_sssp_expand(int32_t const* q, int64_t n, int32_t const* offsets, int32_t const* indices, int32_t* bigq, sssp_state<WT> s,
                                                          int32_t big_deg)
{
  __shared__ wave_queue_storage<2> wqs;
  wqs.init();
  sssp_relax<WT, PK> f{s, wave_queue(wqs, 0, s.q_next, &s.cnt->n_next), wave_queue(wqs, 1, s.far, &s.cnt->n_far)};
  f.begin();
  expand_frontier_mlp(q, n, offsets, indices, bigq, s.cnt, keep_all{}, f, big_deg);
  f.flush();
}
template <typename WT, bool PK = false>
__global__ void __launch_bounds__(TV_BLOCK) k_sssp_expand_big(int32_t const* bigq, int32_t const* offsets, int32_t const* indices, sssp_state<WT> s)
{
  __shared__ wave_queue_storage<2> wqs;
  wqs.init();
  sssp_relax<WT, PK> f{s, wave_queue(wqs, 0, s.q_next, &s.cnt->n_next), wave_queue(wqs, 1, s.far, &s.cnt->n_far)};
  f.begin();
  expand_big_mlp(bigq, offsets, indices, s.cnt, f);
  f.flush();
}

// ---- wide rounds with the filter: TWO STAGES per wavefront (round 6).
// What a wide round waits for is not the number of probes but the CHAIN of a step: neighbour id -> distance probe -> atomicMin -> queue mark, four
// dependent round trips of 1-2 us each, taken by the whole wavefront whenever ONE lane of the step needs them (PMC of k_sssp_expand at RMAT-24: waves
// 76 % in SQ_WAIT_ANY, 1.3 TB/s moved; with the filter dropping 80 % of the probes the round took the same time, profiles/r6g_*).  So the survivors of
// the cheap part (neighbour id, weight, filter bit: streamed or L2-resident) are COMPACTED into a per-wavefront LDS buffer, and the expensive chain runs
// over dense groups of 64 x S2_DU candidates: its round trips are paid per surviving candidate, not per step.
#ifndef CGA_S2_U
#define CGA_S2_U 4
#endif
constexpr int S2_U     = CGA_S2_U;               // edges per lane and step of the cheap stage
constexpr int S2_DU    = 4;                      // candidates per lane and step of the drain
constexpr int S2_DRAIN = 64 * S2_DU;             // buffered candidates that start a drain
constexpr int S2_CAP   = S2_DRAIN + 64 * S2_U;   // (a cheap step adds at most 64 * S2_U)
template <typename WT, bool PK>
struct sssp_cand_storage {
  int32_t v[TV_WAVES][S2_CAP];
  typename dist_bits<WT>::type nd[TV_WAVES][S2_CAP];
  uint32_t lab[PK ? TV_WAVES : 1][PK ? S2_CAP : 1];  // biased parent label (packed words only)
};
template <typename WT, bool PK>
struct sssp_two_stage {
  using B      = dist_bits<WT>;
  using bits_t = typename B::type;
  using R      = sssp_relax<WT, PK>;
  R& f;
  int32_t* cv;
  bits_t* cnd;
  uint32_t* clab;
  uint32_t n{0};  // candidates buffered (wave-uniform)
  // cheap stage for S2_U edges of this lane: positions p[k] of rows whose vertex has distance bits du[k] / biased label lab[k]; live[k]: the edge exists
  __device__ __forceinline__ void step(bool const (&live)[S2_U], eoff_t const (&p)[S2_U], bits_t const (&du)[S2_U], uint32_t const (&lab)[S2_U], int32_t const* indices)
  {
    int const lane = threadIdx.x & 63;
    int32_t v[S2_U];
    WT w[S2_U];
    uint32_t bw[S2_U];
#pragma unroll
    for (int k = 0; k < S2_U; ++k) v[k] = indices[live[k] ? p[k] : 0];
#pragma unroll
    for (int k = 0; k < S2_U; ++k) w[k] = f.s.weights[live[k] ? p[k] : 0];
#pragma unroll
    for (int k = 0; k < S2_U; ++k) bw[k] = f.s.flt.bits ? f.s.flt.bits[(uint32_t)v[k] >> 5] : 0u;
#pragma unroll
    for (int k = 0; k < S2_U; ++k) {
      WT const nd = B::from(du[k]) + w[k];
      bool go     = live[k] & (nd < f.s.cutoff) & !((((bw[k] >> ((uint32_t)v[k] & 31u)) & 1u) != 0u) & (nd >= f.ft));
      if constexpr (PK) go = go & (v[k] != f.s.source);
      uint64_t const m = __ballot(go);
      if (go) {
        uint32_t const at = n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        cv[at]  = v[k];
        cnd[at] = B::to(nd);
        if constexpr (PK) clab[at] = lab[k];
      }
      n += (uint32_t)__popcll(m);
    }
  }
  // the expensive chain over the buffered candidates, newest first (the buffer stays a stack), while at least `min_n` of them wait
  __device__ __forceinline__ void drain(uint32_t min_n)
  {
    int const lane = threadIdx.x & 63;
    while (n > 0 && n >= min_n) {
      uint32_t const cnt = min(n, (uint32_t)(64 * S2_DU)), base = n - cnt;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      int32_t v[S2_DU];
      typename R::cand_t c[S2_DU];
      typename R::tok_t tk[S2_DU];
      typename R::tok2_t tk2[S2_DU];
#pragma unroll
      for (int k = 0; k < S2_DU; ++k) {
        uint32_t const i = base + (uint32_t)lane + 64u * (uint32_t)k;
        bool const has   = i < n;
        v[k]             = has ? cv[i] : -1;
        WT const nd      = B::from(has ? cnd[i] : bits_t(0));
        uint32_t lb      = 0;
        if constexpr (PK) lb = has ? clab[i] : 0u;
        c[k] = f.probe(v[k], nd, lb);
      }
#if defined(CGA_ABL_DRAIN) && CGA_ABL_DRAIN == 1  // timing experiments (WRONG results): the probe only ...
      for (int k = 0; k < S2_DU; ++k) asm volatile("" : : "v"(c[k].pass));
#elif defined(CGA_ABL_DRAIN) && CGA_ABL_DRAIN == 2  // ... probe + atomicMin
      for (int k = 0; k < S2_DU; ++k) tk[k] = f.mid(v[k], c[k]);
      for (int k = 0; k < S2_DU; ++k) asm volatile("" : : "v"(tk[k]));
#elif defined(CGA_ABL_DRAIN) && CGA_ABL_DRAIN == 3  // ... + the queue mark, no append
      for (int k = 0; k < S2_DU; ++k) tk[k] = f.mid(v[k], c[k]);
      for (int k = 0; k < S2_DU; ++k) tk2[k] = f.mid2(v[k], c[k], tk[k]);
      for (int k = 0; k < S2_DU; ++k) asm volatile("" : : "v"(tk2[k]));
#else
#pragma unroll
      for (int k = 0; k < S2_DU; ++k) tk[k] = f.mid(v[k], c[k]);
#pragma unroll
      for (int k = 0; k < S2_DU; ++k) tk2[k] = f.mid2(v[k], c[k], tk[k]);
#pragma unroll
      for (int k = 0; k < S2_DU; ++k) f.post(0, v[k], c[k], tk[k], tk2[k]);
#endif
      n = base;
      __builtin_amdgcn_wave_barrier();
    }
  }
};

template <typename WT, bool PK>
__global__ void __launch_bounds__(TV_BLOCK) k_sssp_expand2(int32_t const* q, int64_t n, int32_t const* offsets, int32_t const* indices, int32_t* bigq, sssp_state<WT> s,
                                                           int32_t big_deg)
{
  using B      = dist_bits<WT>;
  using bits_t = typename B::type;
  __shared__ wave_queue_storage<2> wqs;
  __shared__ sssp_cand_storage<WT, PK> cs;
  __shared__ uint32_t s_scan[TV_WAVES][64];
  __shared__ eoff_t s_beg[TV_WAVES][64];
  __shared__ bits_t s_du[TV_WAVES][64];
  __shared__ uint32_t s_lab[TV_WAVES][64];
  wqs.init();
  sssp_relax<WT, PK> f{s, wave_queue(wqs, 0, s.q_next, &s.cnt->n_next), wave_queue(wqs, 1, s.far, &s.cnt->n_far)};
  f.begin();
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  sssp_two_stage<WT, PK> ts{f, cs.v[wave], cs.nd[wave], PK ? cs.lab[wave] : cs.lab[0]};
  int64_t const gwave  = (int64_t)blockIdx.x * TV_WAVES + wave;
  int64_t const nwaves = (int64_t)gridDim.x * TV_WAVES;
  unsigned long long inspected = 0;
  // A wavefront's 64 frontier entries are taken with a STRIDE (the queue seen as 64 rows of `stride` entries, wavefront j takes column j), not as 64
  // consecutive ones: the hub round's frontier is sorted by distance, i.e. by degree on a power-law graph, and consecutive entries put 64 rows of thousands
  // of edges into the first wavefronts and 64 rows of twenty into the last (a wavefront walks its rows of >= 64 edges one at a time); lane order is still
  // the sorted order inside a wavefront.  (round 6, last session: the hub round of an RMAT-24 traversal 0.9 ms for 7 M edges before)
  int64_t const stride = (n + 63) / 64;
  for (int64_t col = gwave; col < stride; col += nwaves) {
    int64_t const i = (int64_t)lane * stride + col;
    int32_t u = -1, deg = 0;
    eoff_t beg = 0;
    bits_t du  = 0;
    uint32_t lab = 0;
    if (i < n) {
      u   = q[i];
      beg = eoff(offsets, u);
      deg = (int32_t)(eoff(offsets, u + 1) - beg);
      if constexpr (PK) { du = pk_dist_bits(s.pk, u); lab = pk_label(s.labels ? s.labels[u] : u); }
      else du = s.dist[u];
    }
    bool const big = deg >= big_deg;
    if (big) {
      uint32_t const nseg = ((uint32_t)deg + BIG_SEG - 1) / BIG_SEG;
      uint32_t const at   = atomicAdd(&s.cnt->n_big, nseg);
      for (uint32_t sgm = 0; sgm < nseg; ++sgm) { bigq[2 * (at + sgm)] = u; bigq[2 * (at + sgm) + 1] = (int32_t)sgm; }
    }
    // whole-wave rows
    uint64_t mid = __ballot(deg >= 64 && !big);
    while (mid) {
      int const src = __ffsll((unsigned long long)mid) - 1;
      mid &= mid - 1;
      int32_t const d  = __shfl(deg, src);
      eoff_t const b   = (eoff_t)__shfl((int)beg, src);
      bits_t duu;
      if constexpr (sizeof(bits_t) == 4) duu = (bits_t)__shfl((int)du, src);
      else duu = (bits_t)__shfl((unsigned long long)du, src);
      uint32_t const labu = (uint32_t)__shfl((int)lab, src);
      for (int32_t p0 = 0; p0 < d; p0 += 64 * S2_U) {
        bool live[S2_U]; eoff_t pp[S2_U]; bits_t dd[S2_U]; uint32_t ll[S2_U];
#pragma unroll
        for (int k = 0; k < S2_U; ++k) { int32_t const p = p0 + lane + 64 * k; live[k] = p < d; pp[k] = b + (eoff_t)(p < d ? p : 0); dd[k] = duu; ll[k] = labu; }
        ts.step(live, pp, dd, ll, indices);
        ts.drain(S2_DRAIN);
      }
      inspected += (lane == 0) ? (unsigned long long)d : 0ull;
    }
    // flattened small rows
    uint32_t const sd = (deg < 64) ? (uint32_t)deg : 0u;
    uint32_t total;
    uint32_t const ex = wave_excl_scan(sd, lane, &total);
    s_scan[wave][lane] = ex;
    s_beg[wave][lane]  = beg;
    s_du[wave][lane]   = du;
    s_lab[wave][lane]  = lab;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t t0 = 0; t0 < total; t0 += 64 * S2_U) {
      bool live[S2_U]; eoff_t pp[S2_U]; bits_t dd[S2_U]; uint32_t ll[S2_U];
#pragma unroll
      for (int k = 0; k < S2_U; ++k) {
        uint32_t const t = t0 + (uint32_t)lane + 64u * (uint32_t)k;
        live[k] = t < total; pp[k] = 0; dd[k] = 0; ll[k] = 0;
        if (t < total) {
          int lo = 0, hi = 63;
#pragma unroll
          for (int st = 0; st < 6; ++st) {
            int const m = (lo + hi + 1) >> 1;
            if (s_scan[wave][m] <= t) lo = m; else hi = m - 1;
          }
          pp[k] = s_beg[wave][lo] + (t - s_scan[wave][lo]);
          dd[k] = s_du[wave][lo];
          ll[k] = s_lab[wave][lo];
        }
      }
      ts.step(live, pp, dd, ll, indices);
      ts.drain(S2_DRAIN);
    }
    __builtin_amdgcn_wave_barrier();
    inspected += (lane == 0) ? (unsigned long long)total : 0ull;
  }
  ts.drain(1);
  if (lane == 0 && inspected) atomicAdd(&cnt_replica(s.cnt)->edges, inspected);
  f.flush();
}
template <typename WT, bool PK>
__global__ void __launch_bounds__(TV_BLOCK) k_sssp_expand2_big(int32_t const* bigq, int32_t const* offsets, int32_t const* indices, sssp_state<WT> s)
{
  using B      = dist_bits<WT>;
  using bits_t = typename B::type;
  __shared__ wave_queue_storage<2> wqs;
  __shared__ sssp_cand_storage<WT, PK> cs;
  wqs.init();
  sssp_relax<WT, PK> f{s, wave_queue(wqs, 0, s.q_next, &s.cnt->n_next), wave_queue(wqs, 1, s.far, &s.cnt->n_far)};
  f.begin();
  int const wave = threadIdx.x >> 6;
  sssp_two_stage<WT, PK> ts{f, cs.v[wave], cs.nd[wave], PK ? cs.lab[wave] : cs.lab[0]};
  uint32_t const nseg = s.cnt->n_big;
  unsigned long long inspected = 0;
  for (uint32_t k = blockIdx.x; k < nseg; k += gridDim.x) {
    int32_t const u = bigq[2 * k], sgm = bigq[2 * k + 1];
    eoff_t const row_b = eoff(offsets, u), row_e = eoff(offsets, u + 1);
    eoff_t const b = row_b + (eoff_t)sgm * (eoff_t)BIG_SEG;
    eoff_t const len = min(row_e - b, (eoff_t)BIG_SEG);
    bits_t du;
    uint32_t lab = 0;
    if constexpr (PK) { du = pk_dist_bits(s.pk, u); lab = pk_label(s.labels ? s.labels[u] : u); }
    else du = s.dist[u];
    for (eoff_t p0 = 0; p0 < len; p0 += (eoff_t)TV_BLOCK * S2_U) {
      bool live[S2_U]; eoff_t pp[S2_U]; bits_t dd[S2_U]; uint32_t ll[S2_U];
#pragma unroll
      for (int j = 0; j < S2_U; ++j) { eoff_t const p = p0 + threadIdx.x + (eoff_t)j * TV_BLOCK; live[j] = p < len; pp[j] = b + (p < len ? p : 0); dd[j] = du; ll[j] = lab; }
      ts.step(live, pp, dd, ll, indices);
      ts.drain(S2_DRAIN);
    }
    if (threadIdx.x == 0) inspected += (unsigned long long)len;
  }
  ts.drain(1);
  if (threadIdx.x == 0 && inspected) atomicAdd(&cnt_replica(s.cnt)->edges, inspected);
  f.flush();
}

// (A wide frontier rebuilt in VERTEX order -- a sweep over the marks, so that a wavefront's 64 rows are consecutive -- was measured SLOWER, 11.4 -> 13.1 ms:
// consecutive vertices have similar degrees, so a wavefront's rows are all of the slow whole-wave kind at once; profiles/r6j_sssp_ab.txt.)

// A round whose frontier holds a large share of ALL edges (RMAT-24: the two rounds after the hubs relax 259 M and 200 M of the graph's 268 M edges):
// frontier-driven expansion gathers ~200-byte rows at 65-75 G edges/s (0.6 TB/s); the same edges are the whole CSR, so the round STREAMS it instead --
// edge positions in order, 12 bytes each (row id, neighbour, weight: orientation_t::edge_rows is built once per graph), the row's membership
// (mark_near[u] == tag), distance and label read from nearly consecutive places -- and feeds the survivors of the filter to the same two-stage
// drain.  No row search, no deferred rows, no imbalance between rows.
template <typename WT, bool PK>
__global__ void __launch_bounds__(TV_BLOCK) k_sssp_sweep(int32_t const* edge_rows, int32_t const* indices, int64_t ne, uint32_t const* mark_near, uint32_t tag, sssp_state<WT> s,
                                                         unsigned long long* prof)  // CUGRAPH_AMD_SSSP_TRACE=2: per wavefront (ticks, ticks inside drains, drains)
{
  unsigned long long const t_begin = wall_clock64();
  unsigned long long t_drain = 0, n_drain = 0;
  using B      = dist_bits<WT>;
  using bits_t = typename B::type;
  __shared__ wave_queue_storage<2> wqs;
  __shared__ sssp_cand_storage<WT, PK> cs;
  wqs.init();
  sssp_relax<WT, PK> f{s, wave_queue(wqs, 0, s.q_next, &s.cnt->n_next), wave_queue(wqs, 1, s.far, &s.cnt->n_far)};
  f.begin();
  int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  sssp_two_stage<WT, PK> ts{f, cs.v[wave], cs.nd[wave], PK ? cs.lab[wave] : cs.lab[0]};
  int64_t const gwave  = (int64_t)blockIdx.x * TV_WAVES + wave;
  int64_t const nwaves = (int64_t)gridDim.x * TV_WAVES;
  constexpr int64_t STEP = 64 * S2_U;
  unsigned long long inspected = 0;
  // The three streams of step i + 1 are requested while step i is still gathering: they are issued AFTER step i's dependent gathers (marks, then the
  // per-destination words), so the counted waits for those do not wait for the streams behind them (one in-order memory counter), and the HBM round trip
  // of the streams is hidden behind the two L2 round trips of the gathers (positions past ne are clamped to the padded tail: loadable, not live).
  int32_t un[S2_U], vn[S2_U];
  WT wn[S2_U];
  auto request = [&](int64_t p0) {
#pragma unroll
    for (int k = 0; k < S2_U; ++k) {  // streamed once per round: non-temporal, so that the distance words the probes hit stay in the Infinity Cache
      int64_t const p = min(p0 + lane + 64 * k, ne);
      un[k] = __builtin_nontemporal_load(edge_rows + p); vn[k] = __builtin_nontemporal_load(indices + p); wn[k] = __builtin_nontemporal_load(s.weights + p);
    }
  };
  int64_t p0 = gwave * STEP;
  if (p0 < ne) request(p0);
  for (; p0 < ne; p0 += nwaves * STEP) {
    int32_t u[S2_U], v[S2_U];
    WT w[S2_U];
    uint32_t mk[S2_U], bw[S2_U];
    bits_t du[S2_U];
    uint32_t lab[S2_U];
    bool live[S2_U];
#pragma unroll
    for (int k = 0; k < S2_U; ++k) { live[k] = p0 + lane + 64 * k < ne; u[k] = live[k] ? un[k] : 0; v[k] = live[k] ? vn[k] : 0; w[k] = wn[k]; }
#pragma unroll
    for (int k = 0; k < S2_U; ++k) mk[k] = mark_near[u[k]];  // (consecutive positions: a few distinct words per load)
#pragma unroll
    for (int k = 0; k < S2_U; ++k) {
      live[k] = live[k] & (mk[k] == tag);
      du[k] = 0; lab[k] = 0; bw[k] = 0u;
      // everything per-DESTINATION only for the edges of frontier rows: a filter-bit probe is a random L2 access, and the L2 serves ~260 G of those per
      // second -- probing all 268 M positions of an RMAT-24 sweep cost 1 ms of a 1.4 ms kernel whatever the frontier held (profiles/r6n_*)
      if (live[k]) {  // (a row's edges are consecutive: most lanes of a step read the same few words of the row's own state)
        if (s.flt.bits) bw[k] = s.flt.bits[(uint32_t)v[k] >> 5];
        if constexpr (PK) { du[k] = pk_dist_bits(s.pk, u[k]); lab[k] = pk_label(s.labels ? s.labels[u[k]] : u[k]); }
        else du[k] = s.dist[u[k]];
      }
    }
    if (p0 + nwaves * STEP < ne) request(p0 + nwaves * STEP);
    uint32_t n_live = 0;
#pragma unroll
    for (int k = 0; k < S2_U; ++k) {
      WT const nd = B::from(du[k]) + w[k];
      bool go     = live[k] & (nd < s.cutoff) & !((((bw[k] >> ((uint32_t)v[k] & 31u)) & 1u) != 0u) & (nd >= f.ft));
      if constexpr (PK) go = go & (v[k] != s.source);
      n_live += live[k] ? 1u : 0u;
      uint64_t const m = __ballot(go);
      if (go) {
        uint32_t const at = ts.n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        ts.cv[at]  = v[k];
        ts.cnd[at] = B::to(nd);
        if constexpr (PK) ts.clab[at] = lab[k];
      }
      ts.n += (uint32_t)__popcll(m);
    }
    inspected += n_live;
    if (prof && ts.n >= (uint32_t)S2_DRAIN) {
      unsigned long long const t0 = wall_clock64();
      ts.drain(S2_DRAIN);
      t_drain += wall_clock64() - t0; ++n_drain;
    } else ts.drain(S2_DRAIN);
  }
  ts.drain(1);
  if (prof && lane == 0) { prof[3 * gwave] = wall_clock64() - t_begin; prof[3 * gwave + 1] = t_drain; prof[3 * gwave + 2] = n_drain; }
  for (int o = 32; o > 0; o >>= 1) inspected += __shfl_xor(inspected, o);
  if (lane == 0 && inspected) atomicAdd(&cnt_replica(s.cnt)->edges, inspected);
  f.flush();
}
__global__ void k_sssp_edge_rows(int32_t const* offsets, int64_t nv, int32_t* rows)
{  // one wavefront per row chunk (rows[e] = v for offsets[v] <= e < offsets[v + 1]; unsigned positions)
  int64_t const wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int const lane = threadIdx.x & 63;
  for (int64_t v = wave; v < nv; v += nwaves) {
    uint32_t const b = (uint32_t)offsets[v], len = (uint32_t)offsets[v + 1] - b;
    for (uint32_t p = lane; p < len; p += 64) rows[b + p] = (int32_t)v;
  }
}

// ---- the threshold and the bits of sssp_filter.  Distances are cut into SF_BANDS bands of the current window [lower, lower + SF_BANDS / inv_band):
//   hist[0][b]  out-edges of the FRONTIER vertices whose distance lies in band b     (k_sssp_filter_hist_front)
//   hist[1][b]  (1 + out-degree) of ALL vertices whose distance lies in band b       (k_sssp_filter_hist_all; the out-degree stands in for the
//               in-degree: how many edges point at the vertex -- a heuristic, the filter is exact whatever T is)
// sssp_filter_pick maximises, over the band boundaries T, (share of the frontier's relaxations with nd >= T) x (share of the destinations with
// d < T), modelling a relaxation's weight as uniform on (0, 2 avg_w] (constant weights: exactly avg_w).
constexpr int SF_BANDS = 256;
template <typename WT, bool PK>
__device__ __forceinline__ int sf_band(typename dist_bits<WT>::type const* dist, int32_t v, WT lower, WT inv_band)
{
  using B = dist_bits<WT>;
  WT d;
  if constexpr (PK) d = B::from(pk_dist_bits(reinterpret_cast<unsigned long long const*>(dist), v));
  else d = B::from(dist[v]);
  WT const r = (d - lower) * inv_band;
  return r < WT(0) ? 0 : (r >= WT(SF_BANDS) ? SF_BANDS : (int)r);  // SF_BANDS = beyond the window (far pile, unreached): never below a threshold
}
template <typename WT, bool PK>
__global__ void __launch_bounds__(256) k_sssp_filter_hist(int32_t const* front, int64_t n, typename dist_bits<WT>::type const* dist, int32_t const* out_offsets, WT lower,
                                                          WT inv_band, unsigned long long* hist /* [SF_BANDS + 1] */, unsigned long long plus, int sample)
{  // front == nullptr: all vertices 0 .. n - 1.  sample = S: every S-th block of 256 consecutive entries only -- the histograms steer a heuristic (the
   // threshold of an exact filter), a 1 / S sample of a million entries has the same shape, and the two passes were 160 us of every filtered round
  __shared__ unsigned long long h[SF_BANDS + 1];
  for (int i = threadIdx.x; i <= SF_BANDS; i += blockDim.x) h[i] = 0ull;
  __syncthreads();
  for (int64_t blk = blockIdx.x; blk * sample * 256 < n; blk += gridDim.x) {
    int64_t const i = blk * sample * 256 + threadIdx.x;
    if (i >= n) continue;
    int32_t const v = front ? front[i] : (int32_t)i;
    int const b     = sf_band<WT, PK>(dist, v, lower, inv_band);
    if (b < SF_BANDS) atomicAdd(&h[b], plus + (unsigned long long)(eoff(out_offsets, v + 1) - eoff(out_offsets, v)));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < SF_BANDS; i += blockDim.x)
    if (h[i]) atomicAdd(&hist[i], h[i]);
}
template <typename WT>
__global__ void __launch_bounds__(SF_BANDS) k_sssp_filter_pick(unsigned long long* hist_front, unsigned long long* hist_all, WT lower, WT band, WT avg_w, int uniform_weights,
                                                               WT* t_out)
{
  // (round 6, last session: single precision, one reciprocal, a tree for the maximum -- 39 -> a few us per filtered round; the choice steers a heuristic)
  __shared__ float hf[SF_BANDS], ha[SF_BANDS], score[SF_BANDS];
  __shared__ int arg[SF_BANDS];
  int const b = threadIdx.x;
  hf[b] = (float)hist_front[b]; ha[b] = (float)hist_all[b];
  hist_front[b] = 0ull; hist_all[b] = 0ull;  // ready for the next round
  __syncthreads();
  // threshold T = lower + b * band: destinations below it = bands [0, b); a frontier vertex of band c relaxes with nd = (lower + (c + 0.5) band) + w
  float below = 0.0f;
  for (int c = 0; c < b; ++c) below += ha[c];
  float above = 0.0f;
  float const inv2w = 1.0f / (2.0f * (float)avg_w);
  for (int c = 0; c < SF_BANDS; ++c) {
    float const need = ((float)b - ((float)c + 0.5f)) * (float)band;  // nd >= T  <=>  w >= need
    float pw;
    if (need <= 0.0f) pw = 1.0f;
    else if (uniform_weights) pw = (float)avg_w >= need ? 1.0f : 0.0f;
    else pw = fmaxf(0.0f, 1.0f - need * inv2w);
    above += hf[c] * pw;
  }
  score[b] = below * above;
  arg[b]   = b;
  __syncthreads();
  for (int o = SF_BANDS / 2; o > 0; o >>= 1) {  // maximum, the smallest band among equals (as the sequential scan chose)
    if (b < o) {
      float const s1 = score[b + o];
      int const a1   = arg[b + o];
      if (s1 > score[b] || (s1 == score[b] && a1 < arg[b])) { score[b] = s1; arg[b] = a1; }
    }
    __syncthreads();
  }
  if (b == 0) *t_out = lower + (WT)arg[0] * band;  // (band 0: T = lower, nothing is below it: the filter passes everything)
}
template <typename WT, bool PK>
__global__ void __launch_bounds__(256) k_sssp_filter_bits(typename dist_bits<WT>::type const* dist, int64_t nv, WT const* t, uint32_t* bits)
{
  using B = dist_bits<WT>;
  WT const T = *t;
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t const n_pad = (nv + 63) & ~(int64_t)63;
  for (; i < n_pad; i += (int64_t)gridDim.x * blockDim.x) {
    bool below = false;
    if (i < nv) {
      WT d;
      if constexpr (PK) d = B::from(pk_dist_bits(reinterpret_cast<unsigned long long const*>(dist), (int32_t)i));
      else d = B::from(dist[i]);
      below = d < T;
    }
    unsigned long long const m = __ballot(below);
    if ((threadIdx.x & 63) == 0) { bits[i >> 5] = (uint32_t)m; bits[(i >> 5) + 1] = (uint32_t)(m >> 32); }
  }
}

// far pile -> (near frontier | far pile'): d < lower: settled meanwhile, drop; d < upper: near; else keep
template <typename WT, bool PK = false>  // PK: `dist` points at the packed (distance, parent) words
__global__ void __launch_bounds__(TV_BLOCK) k_sssp_split(int32_t const* far_in, int64_t n, typename dist_bits<WT>::type const* dist, WT lower,
                                                         WT upper, int32_t* near_out, int32_t* far_out, uint32_t* mark_near,
                                                         uint32_t* mark_far, uint32_t round, uint32_t new_epoch, counters_t* cnt, int32_t const* out_offsets)
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

// packed (distance, parent) words -> the two result columns; low word kPkNoParent (unreached, or the source) -> -1, else the label without its bias
__global__ void k_sssp_unpack(unsigned long long const* pk, int64_t n, uint32_t* dist_bits_out, int32_t* pred)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned long long const w = pk[i];
    dist_bits_out[i] = (uint32_t)(w >> 32);
    uint32_t const p = (uint32_t)w;
    pred[i]          = p == (uint32_t)kPkNoParent ? -1 : (int32_t)(p ^ 0x80000000u);
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
  // the result's vertex column is a copy of the numbering: 128 MB of traffic at RMAT-24 that nothing in the traversal waits for -- on the side stream
  // (the levels are latency-bound, the copy rides beside them); joined before the result is handed out
  if (nv > 0) hipLaunchKernelGGL(k_copy_i32x4, grid_for(nv / 4 + 1, 256, 1024), 256, 0, h.side_fork(), (int32_t const*)g.number_map.data(), ids->buf.as<int32_t>(), nv);
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
  int32_t const narrow_seg = BIG_SEG_NARROW;  // edges per deferred work unit of a NARROW frontier (bigq is sized for segments of >= 128 edges: big_queue_entries)
  int const bu_grid = (int)std::max<int64_t>(1, std::min<int64_t>(((nv + 63) / 64 + TV_WAVES * BU_GROUPS - 1) / (TV_WAVES * BU_GROUPS),
                                                                  (int64_t)h.num_cus * 16));
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
                             next.data(), dist->buf.as<int32_t>(), pred_p, (int32_t)(depth + 1), cnt.data(), prof.data(), labels);
          unsigned long long pr[8];
          h.read_back(pr, (unsigned long long const*)prof.data(), 8);
          double const nw = (double)bu_grid * TV_WAVES;
          fprintf(stderr, "[bfs bottom-up profile] depth %lld: per wavefront (100 MHz ticks) first chunk %.0f, lane tail %.0f, long rows + write %.0f; slowest wavefront %llu "
                          "(long rows %llu); tail steps %.1f, long rows %.2f, long-row steps %.1f per wavefront\n",
                  (long long)depth, pr[0] / nw, pr[1] / nw, pr[2] / nw, pr[6], pr[7], pr[3] / nw, pr[4] / nw, pr[5] / nw);
        } else {
          hipLaunchKernelGGL(k_bfs_bottom_up<false>, bu_grid, TV_BLOCK, 0, h.stream, in_off, in_idx, out_off, nv, vis_new.data(), (uint32_t const*)front.data(),
                             next.data(), dist->buf.as<int32_t>(), pred_p, (int32_t)(depth + 1), cnt.data(), (unsigned long long*)nullptr, labels);
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
      bfs_state s{dist->buf.as<int32_t>(), pred_p, vis_prev.data(), vis_new.data(), q_nxt,
                  cnt.data(), out_off, in_off, (int32_t)(depth + 1)};
      {
        timed_launch t(h, "bfs_expand");
        int32_t const seg = big_seg_for_deg(big_deg_for(h, n_cur), narrow_seg);
        hipLaunchKernelGGL(k_bfs_expand, expand_grid(h, n_cur), TV_BLOCK, 0, h.stream, (int32_t const*)q_cur, n_cur, (int32_t const*)o.offsets.data(),
                           (int32_t const*)o.indices.data(), bigq.data(), s, big_deg_for(h, n_cur), seg);
        hipLaunchKernelGGL(k_bfs_expand_big, h.num_cus * 8, TV_BLOCK, 0, h.stream, (int32_t const*)bigq.data(), (int32_t const*)o.offsets.data(),
                           (int32_t const*)o.indices.data(), s, seg);
      }
      if (pred_p && labels)  // this level's discoveries: parents as the caller's ids (see k_bfs_relabel_queue)
        hipLaunchKernelGGL(k_bfs_relabel_queue, std::min(h.num_cus, 64), 256, 0, h.stream, (int32_t const*)q_nxt, (counters_t const*)cnt.data(), pred_p, labels);
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
  h.last_stats = cugraph_amd_traversal_stats_t{levels, edges, reached_total, edges_of_reached, 0};
  (void)bu_levels;
  // the predecessors are in the caller's ids already (bfs.cpp:131-138: per level here, not in a pass over V at the end); the vertex column arrives from the side stream
  h.side_join();
  mark("finish");
  h.sync();
  auto* r = new paths_result_t{ids.release(), dist.release(), preds.release()};
  outer_replace_ids(h, g, r->vertex_ids);
  outer_replace_dist(h, g, r->distances);  // BFS distances carry the vertex type (bfs.cpp:156-187)
  outer_replace_ids(h, g, r->predecessors);
  return r;
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
  // The reference's bucket width is 32 * average weight / average degree (sssp_impl.cuh:233-247); the width decides the schedule, never the result
  // (distances and canonical parents are unique).  Here it is FOUR times that: with the streamed wide rounds and the distance filter the cost of a
  // traversal's tail is its number of rounds (19 -> 14 at RMAT-24, relaxations per edge unchanged at 1.47), not the re-relaxations a wider window
  // allows -- 8.6 -> 8.1-8.2 ms over 32 roots for x 4 ... x 64, 10.0 ms at x 1024, 10.3 ... 27.5 ms for x 1/4 ... x 1/32 (profiles/r6av, r6aw)
  double avg_w   = g.ne > 0 ? wsum_h / (double)g.ne : 1.0;
  double avg_deg = nv > 0 ? (double)g.ne / (double)nv : 1.0;
  double delta   = 4.0 * avg_w * 32.0 / std::max(avg_deg, 1.0);
  if (char const* e = getenv("CUGRAPH_AMD_SSSP_DELTA_SCALE")) delta *= atof(e);  // tuning knob (bucket width multiplier)
  if (!(delta > 0.0) || !std::isfinite(delta)) delta = 1.0;
  // Schedules that were built, tested bit for bit against Dijkstra and measured SLOWER than this one over rounds 2-5 are gone from the source
  // (their numbers stay in profiles/ and DESIGN.md section 3.4): light / heavy buckets (1.24 relaxations per edge instead of 2.25 -- and 12.4 ms
  // instead of 11.5: what a round costs is the successful updates and the far-pile re-splits, not the failed relaxations), distance-ordered /
  // radix-heap sub-queues inside the window and device-driven rounds (12.1-21.8 ms), pull rounds over the in-edges for the hub round (12.1 against 11.0).
  int32_t const* row_beg = o.offsets.data();
  int32_t const* adj     = o.indices.data();
  uint64_t steps = 0, relaxed = 0, probed = 0;
  counters_t c;
  {  // d[source] = 0, near = {source}
    bits_t zero_bits = 0;
    HIP_TRY(hipMemcpyAsync(d + source, &zero_bits, sizeof(bits_t), hipMemcpyHostToDevice, h.stream));
    HIP_TRY(hipMemcpyAsync(qa.data(), &source, 4, hipMemcpyHostToDevice, h.stream));
    h.sync();
  }
  int32_t* q_cur = qa.data();
  int32_t* q_nxt = qb.data();
  int32_t* far_cur = fa.data();
  int32_t* far_nxt = fb.data();
  int64_t n_cur = 1, n_far = 0;
  uint32_t round = 0, far_epoch = 1;
  double lower = 0.0, upper = delta;
  uint64_t front_edges = 0;  // out-edges of the current near frontier (0 for the source)
  // fp32 + predecessors: (distance, parent) packed into one 64-bit word per vertex, lowered by one atomicMin per successful relaxation
  // (sssp_relax<WT, true>): no sweep over the settled edges afterwards.  CUGRAPH_AMD_SSSP_PACKED=0 keeps the sweep (what fp64 runs).
  bool packed = false;
  dvec<unsigned long long> pk;
  {
    char const* env_pk = getenv("CUGRAPH_AMD_SSSP_PACKED");
    packed = compute_predecessors && sizeof(WT) == 4 && !(env_pk && std::string(env_pk) == "0");
    if (packed) {
      pk.resize_discard(n1);
      unsigned long long const none = ((unsigned long long)(uint32_t)unreached_bits << 32) | kPkNoParent;
      hipLaunchKernelGGL(k_fill_t<unsigned long long>, grid_for(nv, kBlock, 4096), kBlock, 0, h.stream, pk.data(), nv, none);
      unsigned long long const at_source = kPkNoParent;  // distance 0, no parent
      HIP_TRY(hipMemcpyAsync(pk.data() + source, &at_source, 8, hipMemcpyHostToDevice, h.stream));
      h.sync();  // (at_source is a local)
    }
  }
  bits_t const* const dist_words = packed ? reinterpret_cast<bits_t const*>(pk.data()) : (bits_t const*)d;  // what the PK = packed kernels read distances from
  static bool const sssp_trace = getenv("CUGRAPH_AMD_SSSP_TRACE") != nullptr;  // per round: sizes and wall time since the previous line (stderr)
  auto t_trace = std::chrono::steady_clock::now();
  // The hub round -- the few ten thousand hubs right after the source, whose out-edges are a third of the graph and whose relaxations mostly
  // SUCCEED (a vertex reached from k hubs is lowered ~ln k times when they arrive in any order) -- takes its frontier ordered by tentative
  // distance (256 bands of the window, one 8-bit radix pass): the closest hubs go first, later candidates mostly fail the cheap pre-test.
  // RMAT-24, weights 1..255, 16 roots, same session: 12.19 -> 11.78 ms with predecessors, 11.04 -> 10.74 without (profiles/r5g_sssp_sort.txt);
  // ordering EVERY wide round costs more (the sorts) than the 5 % of relaxations it saves (r5f_sssp_sort.txt).
  bool const sort_hubs = true;
  dvec<uint64_t> sk, sk_out;
  dvec<uint32_t> sv, sv_out, sort_hist;
  // the distance filter (sssp_filter): rounds of at least ne / 32 relaxations (a build costs two passes over the distances, ~60 us at RMAT-24)
  char const* env_flt = getenv("CUGRAPH_AMD_SSSP_FILTER");  // 0: off (the A/B switch of the parity test and of the bench's comparison line)
  bool const filter_force = env_flt && std::string(env_flt) == "force";  // every round, whatever its size (the parity test)
  bool const filter_on = filter_force || (!(env_flt && std::string(env_flt) == "0") && nv >= 4096);
  uint64_t const filter_min_edges = filter_force ? 0 : std::max<uint64_t>((uint64_t)g.ne / 32, (uint64_t)1 << 20);
  // the filter's histograms are built from every 32nd block of 256 entries (k_sssp_filter_hist): RMAT-24, 32 roots: 8.09 ms with every entry, 7.53 with every 8th block,
  // 7.41 with every 32nd (profiles/r6bb_sssp_filter_sample.txt); CUGRAPH_AMD_SSSP_FILTER_SAMPLE=1: every entry
  int sf_sample = 32;
  if (char const* e = getenv("CUGRAPH_AMD_SSSP_FILTER_SAMPLE")) sf_sample = std::max(1, atoi(e));
  char const* env_sw = getenv("CUGRAPH_AMD_SSSP_SWEEP");  // share of the graph's edges a frontier must hold for a streamed round (0: never)
  double const sweep_frac = env_sw ? atof(env_sw) : 0.25;
  bool const sweep_on = sweep_frac > 0.0 && g.ne > 0;
  int const s2_wg_per_cu = 4;  // workgroups per CU of the two-stage kernels (their LDS buffers allow 4 residents; 3: 8.8 -> 9.1 ms, profiles/r6p_*)
  dvec<uint32_t> fbits;
  dvec<unsigned long long> fhist;
  dvec<WT> ft;
  if (filter_on) {
    fbits.resize_discard((size_t)((nv + 63) / 64) * 2 + 2);
    fhist.resize_discard(2 * (SF_BANDS + 1));
    ft.resize_discard(1);
    HIP_TRY(hipMemsetAsync(fhist.data(), 0, 2 * (SF_BANDS + 1) * sizeof(unsigned long long), h.stream));
  }
  auto with_words = [&](auto&& fn) {  // fn(std::bool_constant<PK>): the kernels that read distances exist for the packed and the plain layout
    if constexpr (sizeof(WT) == 4) {
      if (packed) { fn(std::true_type{}); return; }
    }
    fn(std::false_type{});
  };
  auto relax_round = [&](int32_t const* front, int64_t n_front) {
    uint32_t const front_tag = round;  // every member of `front` carries mark_near == the round (or window advance) that appended it
    ++round;
    ++steps;
    WT const lo = (WT)std::min(lower, (double)wmax);
    bool const hub_round = front_edges >= std::max<uint64_t>((uint64_t)g.ne / 10, (uint64_t)1 << 22) && n_front * 16 <= nv;  // few vertices, a large share of the edges
    if (sort_hubs && hub_round && n_front >= 1024 && !g.weights_uniform) {
      if ((int64_t)sk.size() < n_front) {
        sk.resize_discard((size_t)n_front); sk_out.resize_discard((size_t)n_front); sv.resize_discard((size_t)n_front); sv_out.resize_discard((size_t)n_front);
        sort_hist.resize_discard(radix_pass_scratch(n_front));
      }
      WT const inv = (WT)(256.0 / delta);
      with_words([&](auto pkc) {
        hipLaunchKernelGGL((k_sssp_band_keys<WT, decltype(pkc)::value>), grid_for(n_front, kBlock, 4096), kBlock, 0, h.stream, front, n_front, dist_words, lo, inv, sk.data(), sv.data());
      });
      radix_pass_u64_u32(h, sk.data(), sv.data(), sk_out.data(), sv_out.data(), n_front, 0, 8, sort_hist.data());
      front = reinterpret_cast<int32_t const*>(sv_out.data());
    }
    bool const filtered = filter_on && front_edges >= filter_min_edges;
    if (filtered) {
      WT const band = (WT)(delta / SF_BANDS), inv = (WT)(SF_BANDS / delta);
      with_words([&](auto pkc) {
        constexpr bool PKC = decltype(pkc)::value;
        // (at least 2^16 samples: a 1 / 32 sample from 2 M entries on)
        int const s_f = (int)std::clamp<int64_t>(n_front >> 16, 1, sf_sample), s_v = (int)std::clamp<int64_t>(nv >> 16, 1, sf_sample);
        hipLaunchKernelGGL((k_sssp_filter_hist<WT, PKC>), grid_for(n_front / s_f + 1, 256, 1024), 256, 0, h.stream, front, n_front, dist_words, row_beg, lo, inv, fhist.data(), 0ull, s_f);
        hipLaunchKernelGGL((k_sssp_filter_hist<WT, PKC>), grid_for(nv / s_v + 1, 256, 2048), 256, 0, h.stream, (int32_t const*)nullptr, nv, dist_words, row_beg, lo, inv,
                           fhist.data() + (SF_BANDS + 1), 1ull, s_v);
        hipLaunchKernelGGL(k_sssp_filter_pick<WT>, 1, SF_BANDS, 0, h.stream, fhist.data(), fhist.data() + (SF_BANDS + 1), lo, band, (WT)avg_w, g.weights_uniform ? 1 : 0, ft.data());
        hipLaunchKernelGGL((k_sssp_filter_bits<WT, PKC>), grid_for(nv, 256, 2048), 256, 0, h.stream, dist_words, nv, (WT const*)ft.data(), fbits.data());
      });
    }
    counters_t z{};
    z.n_far           = (uint32_t)n_far;
    z.far_min_bits_lo = 0xFFFFFFFFu;
    z.far_min_bits64  = ~0ull;
    std::memcpy(h.pinned, &z, sizeof(z));
    HIP_TRY(hipMemcpyAsync(cnt.data(), h.pinned, sizeof(z), hipMemcpyHostToDevice, h.stream));
    sssp_state<WT> s{d, w, q_nxt, far_cur, mark_near.data(), mark_far.data(), cnt.data(), (WT)std::min(upper, (double)wmax), cutoff, round, far_epoch, row_beg};
    if (packed) { s.pk = pk.data(); s.labels = g.renumbered ? g.number_map.data() : nullptr; s.source = source; }
    if (filtered) { s.flt.bits = fbits.data(); s.flt.t = ft.data(); }
    {
      timed_launch t(h, "sssp_relax");
      with_words([&](auto pkc) {
        constexpr bool PKC = decltype(pkc)::value;
        if (filtered && sweep_on && (double)front_edges >= sweep_frac * (double)g.ne) {  // the frontier's edges are most of the graph: stream the edge list
          orientation_t& ow = g.csr;
          if (ow.edge_rows.size() == 0) {
            ow.edge_rows.resize_discard((size_t)g.ne + kEdgePad);
            HIP_TRY(hipMemsetAsync(ow.edge_rows.data() + g.ne, 0, kEdgePad * sizeof(int32_t), h.stream));
            hipLaunchKernelGGL(k_sssp_edge_rows, grid_for(nv * 16, kBlock, 8192), kBlock, 0, h.stream, row_beg, nv, ow.edge_rows.data());
          }
          static bool const prof_on = getenv("CUGRAPH_AMD_SSSP_TRACE") && atoi(getenv("CUGRAPH_AMD_SSSP_TRACE")) >= 2;
          size_t const n_prof = (size_t)h.num_cus * s2_wg_per_cu * TV_WAVES;
          dvec<unsigned long long> prof(prof_on ? 3 * n_prof : 1);
          hipLaunchKernelGGL((k_sssp_sweep<WT, PKC>), h.num_cus * s2_wg_per_cu, TV_BLOCK, 0, h.stream, (int32_t const*)ow.edge_rows.data(), adj, g.ne, (uint32_t const*)mark_near.data(), front_tag, s,
                             prof_on ? prof.data() : (unsigned long long*)nullptr);
          if (prof_on) {
            std::vector<unsigned long long> pv(3 * n_prof);
            HIP_TRY(hipMemcpyAsync(pv.data(), prof.data(), 3 * n_prof * sizeof(unsigned long long), hipMemcpyDeviceToHost, h.stream));
            h.sync();
            std::vector<unsigned long long> tot, dr;
            double nd = 0;
            for (size_t i = 0; i < n_prof; ++i) { tot.push_back(pv[3 * i]); dr.push_back(pv[3 * i + 1]); nd += (double)pv[3 * i + 2]; }
            std::sort(tot.begin(), tot.end()); std::sort(dr.begin(), dr.end());
            fprintf(stderr, "[sssp sweep] %zu wavefronts: ticks (100 MHz) total min %llu p50 %llu p90 %llu max %llu; inside drains min %llu p50 %llu p90 %llu max %llu; drains per wavefront %.1f\n", n_prof,
                    tot[0], tot[n_prof / 2], tot[n_prof * 9 / 10], tot[n_prof - 1], dr[0], dr[n_prof / 2], dr[n_prof * 9 / 10], dr[n_prof - 1], nd / (double)n_prof);
          }
        } else if (filtered) {  // wide round: survivors of the filter compacted per wavefront, the atomic chain over dense groups (sssp_two_stage)
          hipLaunchKernelGGL((k_sssp_expand2<WT, PKC>), std::min(expand_grid(h, n_front), h.num_cus * s2_wg_per_cu), TV_BLOCK, 0, h.stream, front, n_front, row_beg, adj, bigq.data(), s, big_deg_for(h, n_front));
          hipLaunchKernelGGL((k_sssp_expand2_big<WT, PKC>), h.num_cus * s2_wg_per_cu, TV_BLOCK, 0, h.stream, (int32_t const*)bigq.data(), row_beg, adj, s);
        } else {
          hipLaunchKernelGGL((k_sssp_expand<WT, PKC>), expand_grid(h, n_front), TV_BLOCK, 0, h.stream, front, n_front, row_beg, adj, bigq.data(), s, big_deg_for(h, n_front));
          hipLaunchKernelGGL((k_sssp_expand_big<WT, PKC>), h.num_cus * 8, TV_BLOCK, 0, h.stream, (int32_t const*)bigq.data(), row_beg, adj, s);
        }
      });
    }
    h.read_back(&c, cnt.data(), 1);
    c.fold();
    front_edges = c.out_edges;
    if (sssp_trace) {
      auto const now = std::chrono::steady_clock::now();
      WT t_h = WT(0);
      if (filtered) h.read_back(&t_h, (WT const*)ft.data(), 1);
      fprintf(stderr, "[sssp] round %3u window [%g, %g)  frontier %9lld  edges %11llu  probes %11llu%s  next %9u (%llu out-edges)  far %9u  deferred segments %7u  %8.1f us\n", round,
              lower, upper, (long long)n_front, (unsigned long long)c.edges, (unsigned long long)c.in_edges, filtered ? (" (filter T = " + std::to_string((double)t_h) + ")").c_str() : "",
              c.n_next, (unsigned long long)c.out_edges, c.n_far, c.n_big, std::chrono::duration<double, std::micro>(now - t_trace).count());
      t_trace = std::chrono::steady_clock::now();
    }
    relaxed += c.edges;
    probed += c.in_edges;
    n_cur = c.n_next;
    n_far = c.n_far;
    CGA_EXPECTS(n_far <= nv, CUGRAPH_UNKNOWN_ERROR, "sssp: far pile overflow");
    std::swap(q_cur, q_nxt);
  };
  for (;;) {
    while (n_cur > 0) relax_round(q_cur, n_cur);  // one bucket
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
      with_words([&](auto pkc) {
        hipLaunchKernelGGL((k_sssp_split<WT, decltype(pkc)::value>), grid_for(n_far, TV_BLOCK, 2048), TV_BLOCK, 0, h.stream, (int32_t const*)far_cur, n_far, dist_words,
                           (WT)std::min(lower, (double)wmax), (WT)std::min(upper, (double)wmax), q_cur, far_nxt, mark_near.data(), mark_far.data(), round, far_epoch, cnt.data(), row_beg);
      });
      h.read_back(&c, cnt.data(), 1);
      c.fold();
      front_edges = c.out_edges;
      n_cur = c.n_next;
      n_far = c.n_far;
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
  h.last_stats = cugraph_amd_traversal_stats_t{steps, relaxed, nreached, compute_predecessors ? c.edges : 0, probed};
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
