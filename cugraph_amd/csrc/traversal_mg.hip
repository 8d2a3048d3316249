// Partitioned BFS / SSSP: the per-rank engine behind cugraph_amd/mg_traversal.py (one process per GPU; the all-to-all of
// candidates between the calls below is torch.distributed = RCCL).
//
// Replaces the multi-GPU halves of:
//   detail::bfs / detail::sssp with multi_gpu = true        cpp/src/traversal/bfs_impl.cuh:133-870, sssp_impl.cuh:169-566
//   transform_reduce_if_v_frontier_outgoing_e_by_dst (MG)   prims/transform_reduce_if_v_frontier_outgoing_e_by_dst.cuh:617-1127
//     (local expansion -> (dst, payload) shuffled to the dst owner -> reduced again there)
//   update_v_frontier / fill_edge_dst_property (MG)         prims/update_v_frontier.cuh:164-244, fill_edge_src_dst_property.cuh
//
// Layout.  P ranks; vertices in global degree order are dealt round-robin (position p -> rank p % P, local row p / P),
// L = rows per rank rounded up to a multiple of 64.  A vertex is named by its COMPACT GLOBAL ID g = owner * L + row:
// the owner of g is g / L, and the concatenation of the ranks' L-bit local bitmaps IS the global bitmap, so the
// "who has been visited" knowledge travels as one all-gather of L / 8 bytes per rank and level.  A rank holds the CSR
// of the out-edges of its rows with destinations as compact global ids.
//
// One level / relaxation round on a rank:
//   expand   walk the out-edges of the local frontier (the edge-balanced walk of traversal_common.hpp).  Candidates are
//            REDUCED AT THE SENDER in a table indexed by g (BFS: minimum parent; SSSP: minimum (distance, parent) packed in
//            64 bits) -- a destination is sent at most once per rank and round whatever its in-degree -- then bucketed by
//            owner (counting sort: per-workgroup LDS histograms, one scan, one scatter) into the send buffer;
//   (all-to-all of the buckets: host layer)
//   apply    the owner folds the received candidates into its rows with atomicMin / atomicCAS -- the result does not depend
//            on arrival order -- and builds the next local frontier.
// BFS keeps an exact global visited bitmap (all-gather of the new-frontier bits per level), so only genuinely new vertices
// are ever sent.  SSSP is a frontier Bellman-Ford (a vertex is re-expanded whenever its distance dropped); its fixed point
// is the same as Dijkstra's, so distances are bit-identical to the single-GPU path and to the oracle.
// Parents: the minimum EXTERNAL id among the valid parents (BFS: in-neighbours one level up; SSSP: tight in-edges) --
// deterministic and independent of P and of the numbering.
#include "common.hpp"
#include "traversal_common.hpp"
#include "traversal_bottom_up.hpp"

#include <cfloat>
#include <climits>
#include <cstring>
#include <vector>

#include "cugraph_amd/extensions.h"

namespace cga {

namespace {

constexpr int MG_MAX_RANKS = 64;
constexpr int MG_BUCKET_BLOCKS = 512;
constexpr unsigned long long MG_NONE64 = ~0ull;

struct mg_bfs_state {
  uint32_t const* seen;   // global bitmap: visited as of the start of the level
  uint32_t* touched;      // global bitmap: candidates of this level (first setter lists the vertex)
  int32_t* cand_parent;   // [P * L] minimum external id of a frontier parent, INT32_MAX = none
  int32_t const* row_vertex;
  int32_t* cand;          // candidate list (compact global ids)
  counters_t* cnt;
  int with_pred;
};

struct mg_bfs_visit {
  mg_bfs_state s;
  wave_queue wq;
  __device__ __forceinline__ void operator()(int32_t u, int32_t g, eoff_t)
  {
    uint32_t const bit = 1u << (g & 31);
    bool fresh = false;
    if (!(s.seen[g >> 5] & bit)) {
      if (s.with_pred) {
        int32_t const pu = s.row_vertex[u];
        if (pu < __hip_atomic_load(&s.cand_parent[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&s.cand_parent[g], pu);
      }
      if (!(__hip_atomic_load(&s.touched[g >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) fresh = !(atomicOr(&s.touched[g >> 5], bit) & bit);
    }
    wq.push(fresh, g);
  }
};

struct mg_sssp_state {
  unsigned long long* st;        // [n_rows] (distance bits << 32) | (parent external id + 1)
  unsigned long long* cand_best; // [P * L] best candidate of this round per destination
  uint32_t* touched;
  float const* weights;
  int32_t const* row_vertex;
  int32_t* cand;
  counters_t* cnt;
  float cutoff;
  // destinations this rank owns are relaxed IN PLACE (see mg_sssp_relax::pre): their range of compact ids, the round's dedup mark and the
  // next local frontier (appends counted in cnt->n_far; apply continues behind them)
  uint32_t own_lo, own_n;        // compact ids [own_lo, own_lo + own_n) = this rank's rows (own_n = 0: no shortcut)
  uint32_t* mark;
  int32_t* q_next;
  uint32_t round;
  // near / far (round 4): an improved row whose distance is below the window's upper bound joins the next frontier, one beyond it the
  // FAR pile (once per window: far_mark), which cugraph_amd_traversal_mg_plan_sssp_advance splits when the window moves on.  hi_bits =
  // the bound's float bits (non-negative floats order like their bit patterns); 0x7f7fffff (FLT_MAX): no window, everything is near
  uint32_t hi_bits, win;
  uint32_t* far_mark;
  int32_t* q_far;
  uint32_t* far_count;
};

struct mg_sssp_relax {
  mg_sssp_state s;
  wave_queue wq, wq_own, wq_far;
  // the phased form (expand_*_mlp of traversal_common.hpp: EX_U edges in flight per lane).  A candidate whose destination THIS rank owns
  // is applied in place -- atomicMin on st[row], round mark, append to the next local frontier -- instead of going through the candidate
  // table, the bucket sort, the (self-)exchange and apply: an improvement is then visible to the rest of the round (Gauss-Seidel) where the
  // exchanged ones become visible a round later (Jacobi).  With one rank that is every candidate: six sweeps over the edges become
  // three (profiles/r3s_mgsssp_debug.log); with P ranks it is 1 / P of them.  The fixed point -- and the minimum-external-id parent
  // among the tight in-edges -- does not depend on the schedule.
  struct cand_t { unsigned long long packed; uint32_t bit; bool pass, claimed, own; };
  using tok_t  = uint32_t;
  using tok2_t = uint32_t;
  __device__ __forceinline__ cand_t pre(int32_t u, int32_t g, eoff_t p) const
  {  // branch-free
    int32_t const us = u < 0 ? 0 : u, gs = g < 0 ? 0 : g;
    float const nd   = __uint_as_float((uint32_t)(s.st[us] >> 32)) + s.weights[p];
    unsigned long long const packed = ((unsigned long long)__float_as_uint(nd) << 32) | (uint32_t)(s.row_vertex[us] + 1);
    bool const own = ((uint32_t)gs - s.own_lo) < s.own_n;
    unsigned long long const* const cur = own ? &s.st[(uint32_t)gs - s.own_lo] : &s.cand_best[gs];
    unsigned long long const best = __hip_atomic_load(cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t const bit = 1u << (gs & 31);
    bool const claimed = (__hip_atomic_load(&s.touched[gs >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) != 0;
    bool const pass    = (g >= 0) & (nd < s.cutoff) & (packed < best);
    return cand_t{packed, bit, pass, claimed, own};
  }
  __device__ __forceinline__ tok_t mid(int32_t g, cand_t c) const
  {  // -> 1: an owned destination's DISTANCE dropped (it must be expanded again)
    if (!c.pass) return 0u;
    if (c.own) {
      unsigned long long const old = atomicMin(&s.st[(uint32_t)g - s.own_lo], c.packed);
      return (uint32_t)(c.packed >> 32) < (uint32_t)(old >> 32) ? 1u : 0u;
    }
    atomicMin(&s.cand_best[g], c.packed);
    return 0u;
  }
  __device__ __forceinline__ tok2_t mid2(int32_t g, cand_t c, tok_t dropped) const
  {  // -> 1: list g as a candidate for its owner; 2: append the owned row to the next local frontier; 3: to the far pile
    if (!c.pass) return 0u;
    if (c.own) {
      if (!dropped) return 0u;
      uint32_t const row = (uint32_t)g - s.own_lo;
      if ((uint32_t)(c.packed >> 32) < s.hi_bits) return atomicExch(&s.mark[row], s.round) != s.round ? 2u : 0u;
      return atomicExch(&s.far_mark[row], s.win) != s.win ? 3u : 0u;
    }
    return !c.claimed ? (uint32_t)!(atomicOr(&s.touched[g >> 5], c.bit) & c.bit) : 0u;
  }
  __device__ __forceinline__ void post(int32_t, int32_t g, cand_t, tok_t, tok2_t what)
  {
    wq.push(what == 1u, g);
    wq_own.push(what == 2u, (int32_t)((uint32_t)g - s.own_lo));
    wq_far.push(what == 3u, (int32_t)((uint32_t)g - s.own_lo));
  }
};

struct keep_all_mg { __device__ __forceinline__ bool operator()(int32_t) const { return true; } };

__global__ void __launch_bounds__(TV_BLOCK) k_mg_bfs_expand(int32_t const* q, int64_t n, int32_t const* offsets, int32_t const* indices, int32_t* bigq,
                                                            mg_bfs_state s)
{
  __shared__ wave_queue_storage<1> wqs;
  wqs.init();
  mg_bfs_visit f{s, wave_queue(wqs, 0, s.cand, &s.cnt->n_next)};
  expand_frontier(q, n, offsets, indices, bigq, s.cnt, keep_all_mg{}, f);
  f.wq.flush();
}
__global__ void __launch_bounds__(TV_BLOCK) k_mg_bfs_expand_big(int32_t const* bigq, int32_t const* offsets, int32_t const* indices, mg_bfs_state s)
{
  __shared__ wave_queue_storage<1> wqs;
  wqs.init();
  mg_bfs_visit f{s, wave_queue(wqs, 0, s.cand, &s.cnt->n_next)};
  expand_big(bigq, offsets, indices, s.cnt, f);
  f.wq.flush();
}
__global__ void __launch_bounds__(TV_BLOCK) k_mg_sssp_expand(int32_t const* q, int64_t n, int32_t const* offsets, int32_t const* indices, int32_t* bigq,
                                                             mg_sssp_state s)
{
  __shared__ wave_queue_storage<3> wqs;
  wqs.init();
  mg_sssp_relax f{s, wave_queue(wqs, 0, s.cand, &s.cnt->n_next), wave_queue(wqs, 1, s.q_next, &s.cnt->n_far), wave_queue(wqs, 2, s.q_far, s.far_count)};
  expand_frontier_mlp(q, n, offsets, indices, bigq, s.cnt, keep_all_mg{}, f);
  f.wq.flush();
  f.wq_own.flush();
  f.wq_far.flush();
}
__global__ void __launch_bounds__(TV_BLOCK) k_mg_sssp_expand_big(int32_t const* bigq, int32_t const* offsets, int32_t const* indices, mg_sssp_state s)
{
  __shared__ wave_queue_storage<3> wqs;
  wqs.init();
  mg_sssp_relax f{s, wave_queue(wqs, 0, s.cand, &s.cnt->n_next), wave_queue(wqs, 1, s.q_next, &s.cnt->n_far), wave_queue(wqs, 2, s.q_far, s.far_count)};
  expand_big_mlp(bigq, offsets, indices, s.cnt, f);
  f.wq.flush();
  f.wq_own.flush();
  f.wq_far.flush();
}

// ---- bucketing by owner: counting sort of the candidate list.  Workgroup b owns the contiguous slice
// [b * chunk, (b + 1) * chunk) of the list in both passes, chunk = ceil(n / MG_BUCKET_BLOCKS) with n read on the device.
__global__ void __launch_bounds__(256) k_mg_bucket_count(int32_t const* cand, counters_t const* cnt, uint32_t L, int P, uint32_t* block_hist)
{
  __shared__ uint32_t h[MG_MAX_RANKS];
  if (threadIdx.x < MG_MAX_RANKS) h[threadIdx.x] = 0;
  __syncthreads();
  uint32_t const n = cnt->n_next, chunk = (n + gridDim.x - 1) / gridDim.x;
  uint32_t const b = blockIdx.x * chunk, e = min(n, b + chunk);
  for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) atomicAdd(&h[(uint32_t)cand[i] / L], 1u);
  __syncthreads();
  if ((int)threadIdx.x < P) block_hist[threadIdx.x * gridDim.x + blockIdx.x] = h[threadIdx.x];  // owner-major
}

// exclusive scan of the owner-major histogram (P * nb entries, one workgroup) + per-owner totals
__global__ void __launch_bounds__(1024) k_mg_bucket_scan(uint32_t* block_hist, int P, int nb, unsigned long long* totals)
{
  __shared__ uint32_t part[1024];
  int const n = P * nb, per = (n + 1023) / 1024;
  int const b = threadIdx.x * per, e = min(n, b + per);
  uint32_t s = 0;
  for (int i = b; i < e; ++i) s += block_hist[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int i = 0; i < 1024; ++i) { uint32_t t = part[i]; part[i] = run; run += t; }
  }
  __syncthreads();
  uint32_t run = part[threadIdx.x];
  for (int i = b; i < e; ++i) { uint32_t t = block_hist[i]; block_hist[i] = run; run += t; }
  __syncthreads();
  if ((int)threadIdx.x < P) {
    uint32_t const first = block_hist[threadIdx.x * nb];
    uint32_t const next  = (int)threadIdx.x + 1 < P ? block_hist[(threadIdx.x + 1) * nb] : 0xFFFFFFFFu;
    totals[threadIdx.x]  = (unsigned long long)first | ((unsigned long long)next << 32);  // (start, start of the next owner); the last one is fixed on the host
  }
}

// scatter of (row, payload) tuples into the owner buckets; resets the sender-side tables for the next round
template <int MODE>  // 0 = BFS: (row, parent); 1 = SSSP: (row, distance bits, parent)
__global__ void __launch_bounds__(256) k_mg_bucket_scatter(int32_t const* cand, counters_t const* cnt, uint32_t L, int P, uint32_t const* block_base,
                                                           int32_t* cand_parent, unsigned long long* cand_best, uint32_t* touched, int32_t* out)
{
  __shared__ uint32_t cur[MG_MAX_RANKS];
  if ((int)threadIdx.x < P) cur[threadIdx.x] = block_base[threadIdx.x * gridDim.x + blockIdx.x];
  __syncthreads();
  uint32_t const n = cnt->n_next, chunk = (n + gridDim.x - 1) / gridDim.x;
  uint32_t const b = blockIdx.x * chunk, e = min(n, b + chunk);
  for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) {
    uint32_t const g = (uint32_t)cand[i], o = g / L;
    uint32_t const at = atomicAdd(&cur[o], 1u);
    if constexpr (MODE == 0) {
      out[2 * (size_t)at]     = (int32_t)(g - o * L);
      out[2 * (size_t)at + 1] = cand_parent[g];
      cand_parent[g]          = INT32_MAX;
    } else {
      unsigned long long const v = cand_best[g];
      out[3 * (size_t)at]     = (int32_t)(g - o * L);
      out[3 * (size_t)at + 1] = (int32_t)(uint32_t)(v >> 32);
      out[3 * (size_t)at + 2] = (int32_t)(uint32_t)v;
      cand_best[g]            = MG_NONE64;
    }
    touched[g >> 5] = 0;  // every set bit of the word belongs to a listed candidate: plain stores of zero suffice
  }
}

// ---- owner side
// (n_dev != nullptr: the device-driven exchange of traversal_mg_driver.hip, where no host knows the counts: sender s wrote n_dev[s] tuples into ITS slot of the
// window, seg_tuples tuples apart; blockIdx.y = sender, the grid is fixed and every workgroup strides over its sender's list -- trip counts are uniform per
// workgroup, so the wavefront-wide appends stay whole)
__global__ void k_mg_bfs_apply(int32_t const* in, size_t n_host, uint32_t const* n_dev, size_t seg_tuples, int32_t level, int32_t* dist, int32_t* pred, int32_t* q_next,
                               uint32_t* newfront, counters_t* cnt, int32_t const* out_offsets, int32_t const* in_offsets /* nullptr: no degree sums (no bottom-up levels) */)
{
  size_t const n = n_dev ? (size_t)n_dev[blockIdx.y] : n_host;
  in += 2 * seg_tuples * blockIdx.y;
  unsigned long long acc_out = 0, acc_in = 0;
  for (size_t base = blockIdx.x * (size_t)blockDim.x; base < n; base += (size_t)gridDim.x * blockDim.x) {
    size_t const i = base + threadIdx.x;
    bool fresh     = false;
    int32_t row    = 0;
    if (i < n) {
      row                = in[2 * i];
      int32_t const par  = in[2 * i + 1];
      int32_t const old  = atomicCAS(&dist[row], INT32_MAX, level);
      fresh              = old == INT32_MAX;
      if (fresh) {
        atomicOr(&newfront[row >> 5], 1u << (row & 31));
        if (in_offsets) {  // the direction heuristic's sums (bfs_impl.cuh:598-607 counts the same two quantities)
          acc_out += (unsigned long long)(eoff(out_offsets, row + 1) - eoff(out_offsets, row));
          acc_in  += (unsigned long long)(eoff(in_offsets, row + 1) - eoff(in_offsets, row));
        }
      }
      if (pred && (fresh || old == level) && par < __hip_atomic_load(&pred[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&pred[row], par);
    }
    wave_push(fresh, row, q_next, &cnt->n_next, threadIdx.x & 63);
  }
  if (in_offsets) {
    for (int o = 32; o > 0; o >>= 1) { acc_out += __shfl_xor(acc_out, o); acc_in += __shfl_xor(acc_in, o); }
    if ((threadIdx.x & 63) == 0 && (acc_out | acc_in)) { counter_sums_t* r = cnt_replica(cnt); atomicAdd(&r->out_edges, acc_out); atomicAdd(&r->in_edges, acc_in); }
  }
}

__global__ void k_mg_sssp_apply(int32_t const* in, size_t n, uint32_t round, unsigned long long* st, uint32_t* mark, int32_t* q_next, counters_t* cnt,
                                uint32_t hi_bits, uint32_t win, uint32_t* far_mark, int32_t* q_far, uint32_t* far_count)
{
  size_t i    = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  bool push = false, far = false;
  int32_t row = 0;
  if (i < n) {
    row = in[3 * i];
    unsigned long long const packed = ((unsigned long long)(uint32_t)in[3 * i + 1] << 32) | (uint32_t)in[3 * i + 2];
    if (packed < __hip_atomic_load(&st[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
      unsigned long long const old = atomicMin(&st[row], packed);
      if ((uint32_t)(packed >> 32) < (uint32_t)(old >> 32)) {  // the DISTANCE dropped: expand again -- in this window, or when the window reaches it
        if ((uint32_t)(packed >> 32) < hi_bits) push = atomicExch(&mark[row], round) != round;
        else far = atomicExch(&far_mark[row], win) != win;
      }
    }
  }
  wave_push(push, row, q_next, &cnt->n_next, threadIdx.x & 63);
  wave_push(far, row, q_far, far_count, threadIdx.x & 63);
}

// the far pile when the window moves from [.., hi_old) to [.., hi_new): rows whose distance has dropped below hi_old were expanded already;
// rows below hi_new form the next frontier; the rest stay (marked with the new window).  *dmin: smallest distance bits left in the pile.
__global__ void k_mg_sssp_split(int32_t const* far, uint32_t n, unsigned long long const* st, uint32_t hi_old, uint32_t hi_new, uint32_t round, uint32_t win_new,
                                uint32_t* mark, uint32_t* far_mark, int32_t* q_front, uint32_t* n_front, int32_t* far_out, uint32_t* n_far_out)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  bool near = false, keep = false;
  int32_t row = 0;
  if (i < n) {
    row = far[i];
    uint32_t const d = (uint32_t)(st[row] >> 32);
    if (d >= hi_old) {
      if (d < hi_new) near = atomicExch(&mark[row], round) != round;
      else keep = atomicExch(&far_mark[row], win_new) != win_new;
    }
  }
  wave_push(near, row, q_front, n_front, threadIdx.x & 63);
  wave_push(keep, row, far_out, n_far_out, threadIdx.x & 63);
}
__global__ void k_mg_sssp_far_min(int32_t const* far, uint32_t n, unsigned long long const* st, uint32_t hi, uint32_t* out /*[0] live entries, [1] min bits*/)
{
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t d = 0xFFFFFFFFu;
  if (i < n) { uint32_t const x = (uint32_t)(st[far[i]] >> 32); if (x >= hi) d = x; }
  uint32_t live = d != 0xFFFFFFFFu ? 1u : 0u;
  for (int o = 32; o; o >>= 1) { d = min(d, (uint32_t)__shfl_xor((int)d, o)); live += (uint32_t)__shfl_xor((int)live, o); }
  if ((threadIdx.x & 63) == 0 && live) { atomicAdd(&out[0], live); atomicMin(&out[1], d); }
}

__global__ void k_mg_bfs_sources(int32_t const* rows, size_t n, int32_t* dist, int32_t* q, uint32_t* newfront, counters_t* cnt)
{
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t const r = rows[i];
  if (atomicCAS(&dist[r], INT32_MAX, 0) == INT32_MAX) {  // duplicates in the source list are enqueued once
    atomicOr(&newfront[r >> 5], 1u << (r & 31));
    q[atomicAdd(&cnt->n_next, 1u)] = r;
  }
}
__global__ void k_mg_sssp_sources(int32_t const* rows, size_t n, unsigned long long* st, int32_t* q, counters_t* cnt)
{
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t const r = rows[i];
  if (atomicExch(&st[r], 0ull) != 0ull) q[atomicAdd(&cnt->n_next, 1u)] = r;  // (distance 0, no parent): the smallest key, nothing overrides it
}

__global__ void k_mg_or(uint32_t* a, uint32_t const* b, size_t n)
{
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) a[i] |= b[i];
}
template <typename T>
__global__ void k_mg_fill(T* p, size_t n, T v)
{
  size_t i      = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}
__global__ void k_mg_bfs_results(int32_t const* dist, int32_t const* pred, size_t n, int32_t* dist_out, int32_t* pred_out)
{
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  dist_out[i] = dist[i];
  if (pred_out) pred_out[i] = (pred && pred[i] != INT32_MAX) ? pred[i] : -1;
}
__global__ void k_mg_sssp_results(unsigned long long const* st, size_t n, float* dist_out, int32_t* pred_out)
{
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long const v = st[i];
  bool const reached = v != MG_NONE64;
  dist_out[i] = reached ? __uint_as_float((uint32_t)(v >> 32)) : FLT_MAX;  // unreached = type max (sssp_impl.cuh)
  if (pred_out) pred_out[i] = reached ? (int32_t)(uint32_t)v - 1 : -1;  // low word = parent + 1; 0 (-> -1) for the sources
}

inline uint32_t float_bits(float f)
{
  uint32_t b;
  std::memcpy(&b, &f, 4);
  return b;
}

template <typename T>
void fill(handle_t const& h, T* p, size_t n, T v)
{
  if (n) hipLaunchKernelGGL(k_mg_fill<T>, grid_for((int64_t)n, kBlock, 8192), kBlock, 0, h.stream, p, n, v);
}

struct traversal_mg_plan {
  handle_t const* h{nullptr};
  bool caller_keeps_buffers{false};  // the buffers handed to merge_visited / apply outlive the call (the library's own driver: persistent windows)
  int mode{0}, rank{0}, P{1};
  size_t n_rows{0}, n_edges{0}, L{0}, capacity{0};
  int32_t const* offsets{nullptr};
  int32_t const* indices{nullptr};
  float const* weights{nullptr};
  int32_t const* row_vertex{nullptr};
  int32_t* send{nullptr};
  bool with_pred{true};
  float cutoff{FLT_MAX};
  // state
  dvec<int32_t> dist, pred, cand_parent, cand, q_a, q_b, bigq;
  dvec<unsigned long long> st, cand_best, totals;
  dvec<uint32_t> seen, touched, newfront, mark, block_hist;
  dvec<counters_t> cnt;
  int32_t* q_cur{nullptr};
  int32_t* q_next{nullptr};
  size_t n_frontier{0};
  // bottom-up levels (BFS): the in-edges of the local rows (neighbours = compact global ids, ascending EXTERNAL id inside a row) and the
  // external id of every compact global id; degree sums of the last level's discoveries for the direction heuristic
  int32_t const* in_offsets{nullptr};
  int32_t const* in_indices{nullptr};
  int32_t const* ext_of_g{nullptr};
  unsigned long long last_out{0}, last_in{0};
  // SSSP: the relaxation round (dedup mark of expand's in-place relaxations and of apply) and the length of the next local frontier
  // that expand has already written
  uint32_t round{0};
  size_t n_local{0};
  bool own_in_place{true};
  // SSSP near / far window (cugraph_amd_traversal_mg_plan_sssp_set_window / _far_stats / _advance); hi = FLT_MAX: no window
  float hi{FLT_MAX};
  uint32_t win{1};
  dvec<uint32_t> far_mark, far_count;
  dvec<int32_t> q_far, q_far2;
  int tuple_words() const { return mode == 0 ? 2 : 3; }
};

}  // namespace
}  // namespace cga

using namespace cga;

// every entry point goes through here: the plan's stream is named to the memory pool, so blocks freed / reused by the call are ordered on the
// stream its kernels run on (the handle may have borrowed another stream since the plan was created)
static traversal_mg_plan& TP(cugraph_amd_traversal_mg_plan_t* p)
{
  auto& plan = *reinterpret_cast<traversal_mg_plan*>(p);
  if (plan.h) pool_set_stream(plan.h->stream);
  return plan;
}

extern "C" cugraph_error_code_t cugraph_amd_traversal_mg_plan_create(const cugraph_resource_handle_t* handle, const int32_t* offsets,
                                                                     const int32_t* indices, const float* weights, size_t n_rows, size_t n_edges,
                                                                     size_t rows_per_rank, int comm_rank, int comm_size,
                                                                     const int32_t* row_vertex, int mode, int32_t* send, size_t capacity_tuples,
                                                                     cugraph_amd_traversal_mg_plan_t** plan, cugraph_error_t** error)
{
  if (plan) *plan = nullptr;
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    CGA_EXPECTS(plan != nullptr && offsets != nullptr && row_vertex != nullptr && send != nullptr, CUGRAPH_INVALID_INPUT, "NULL argument");
    CGA_EXPECTS(mode == 0 || mode == 1, CUGRAPH_INVALID_INPUT, "mode must be 0 (BFS) or 1 (SSSP)");
    CGA_EXPECTS(mode == 0 || weights != nullptr, CUGRAPH_INVALID_INPUT, "SSSP needs (float) weights");
    CGA_EXPECTS(comm_size >= 1 && comm_size <= MG_MAX_RANKS && comm_rank >= 0 && comm_rank < comm_size, CUGRAPH_INVALID_INPUT, "bad rank / size");
    CGA_EXPECTS(rows_per_rank % 64 == 0 && n_rows <= rows_per_rank, CUGRAPH_INVALID_INPUT, "rows_per_rank must be a multiple of 64 and >= n_rows");
    CGA_EXPECTS((uint64_t)rows_per_rank * (uint64_t)comm_size < ((uint64_t)1 << 31), CUGRAPH_INVALID_INPUT, "compact global ids must fit 31 bits");
    HIP_TRY(hipSetDevice(h.device));
    auto p       = std::make_unique<traversal_mg_plan>();
    p->h         = &h;
    p->mode      = mode;
    p->rank      = comm_rank;
    p->P         = comm_size;
    p->n_rows    = n_rows;
    p->n_edges   = n_edges;
    p->L         = rows_per_rank;
    p->capacity  = capacity_tuples;
    p->offsets   = offsets;
    p->indices   = indices;
    p->weights   = weights;
    p->row_vertex = row_vertex;
    p->send      = send;
    size_t const G = rows_per_rank * (size_t)comm_size, n1 = std::max<size_t>(n_rows, 1);
    CGA_EXPECTS(capacity_tuples >= std::min<size_t>(G, std::max<size_t>(n_edges, 1)), CUGRAPH_INVALID_INPUT,
                "send capacity must hold min(P * L, local edges) tuples");
    p->cand.resize_discard(capacity_tuples + 64);
    p->q_a.resize_discard(n1);
    p->q_b.resize_discard(n1);
    p->bigq.resize_discard(big_queue_entries((int64_t)n_edges));
    p->touched.resize_discard(G / 32);
    p->block_hist.resize_discard((size_t)MG_MAX_RANKS * MG_BUCKET_BLOCKS);
    p->totals.resize_discard(MG_MAX_RANKS);
    p->cnt.resize_discard(1);
    if (mode == 0) {
      p->dist.resize_discard(n1);
      p->pred.resize_discard(n1);
      p->cand_parent.resize_discard(G);
      p->seen.resize_discard(G / 32);
      p->newfront.resize_discard(rows_per_rank / 32);
    } else {
      p->st.resize_discard(n1);
      p->cand_best.resize_discard(G);
      p->mark.resize_discard(n1);
      p->far_mark.resize_discard(n1);
      p->q_far.resize_discard(n1);
      p->q_far2.resize_discard(n1);
      p->far_count.resize_discard(2);
    }
    h.sync();
    *plan = reinterpret_cast<cugraph_amd_traversal_mg_plan_t*>(p.release());
  });
}

extern "C" void cugraph_amd_traversal_mg_plan_keep_buffers(cugraph_amd_traversal_mg_plan_t* plan, bool_t on)
{
  if (plan) TP(plan).caller_keeps_buffers = on == TRUE;
}

// The plan queues its work on -- and reads counters back through -- a resource handle; a caller that creates a handle per call (the reference accepts any
// handle of the communicator) moves a cached plan to the handle of the current call.  Everything the previous handle queued has finished by then (every
// stepping entry point that yields host-visible data ends synchronised), and the device is synchronised here in case a caller overlapped calls anyway.
extern "C" void cugraph_amd_traversal_mg_plan_rebind(cugraph_amd_traversal_mg_plan_t* plan, const cugraph_resource_handle_t* handle)
{
  if (!plan || !handle) return;
  handle_t const* h = reinterpret_cast<handle_t const*>(handle);
  auto& pl = *reinterpret_cast<traversal_mg_plan*>(plan);  // (not TP(): that names the OLD handle's stream to the pool, and the old handle may be gone)
  if (pl.h == h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  pl.h = h;
  pool_set_stream(h->stream);
}

extern "C" void cugraph_amd_traversal_mg_plan_free(cugraph_amd_traversal_mg_plan_t* plan)
{
  if (!plan) return;
  // (not TP(): the handle of the plan's last call may be gone when the graph that caches the plan is freed.  The device is synchronised and the blocks go
  // back to the pool without a stream to order against)
  (void)hipDeviceSynchronize();
  pool_forget_stream(nullptr);
  delete reinterpret_cast<traversal_mg_plan*>(plan);
}

extern "C" cugraph_error_code_t cugraph_amd_traversal_mg_plan_reset(cugraph_amd_traversal_mg_plan_t* plan, const int32_t* source_rows,
                                                                    size_t n_sources, double cutoff, bool_t compute_predecessors,
                                                                    cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(plan != nullptr, CUGRAPH_INVALID_INPUT, "plan is NULL");
    traversal_mg_plan& p = TP(plan);
    handle_t const& h    = *p.h;
    HIP_TRY(hipSetDevice(h.device));
    size_t const G = p.L * (size_t)p.P, n1 = std::max<size_t>(p.n_rows, 1);
    p.with_pred = compute_predecessors == TRUE;
    p.cutoff    = cutoff >= (double)FLT_MAX ? FLT_MAX : (float)cutoff;
    p.round     = 0;
    p.n_local   = 0;
    {
      char const* e  = getenv("CUGRAPH_AMD_MG_SSSP_INPLACE");  // 0: every candidate through the exchange (A/B, tests)
      p.own_in_place = !(e && atoi(e) == 0);
    }
    HIP_TRY(hipMemsetAsync(p.touched.data(), 0, G / 8, h.stream));
    HIP_TRY(hipMemsetAsync(p.cnt.data(), 0, sizeof(counters_t), h.stream));
    p.q_cur  = p.q_a.data();
    p.q_next = p.q_b.data();
    int const g = (int)((n_sources + 255) / 256);
    if (p.mode == 0) {
      fill<int32_t>(h, p.dist.data(), n1, INT32_MAX);
      fill<int32_t>(h, p.pred.data(), n1, INT32_MAX);
      fill<int32_t>(h, p.cand_parent.data(), G, INT32_MAX);
      HIP_TRY(hipMemsetAsync(p.seen.data(), 0, G / 8, h.stream));
      HIP_TRY(hipMemsetAsync(p.newfront.data(), 0, p.L / 8, h.stream));
      if (n_sources) hipLaunchKernelGGL(k_mg_bfs_sources, g, 256, 0, h.stream, source_rows, n_sources, p.dist.data(), p.q_cur, p.newfront.data(), p.cnt.data());
    } else {
      fill<unsigned long long>(h, p.st.data(), n1, MG_NONE64);
      fill<unsigned long long>(h, p.cand_best.data(), G, MG_NONE64);
      HIP_TRY(hipMemsetAsync(p.mark.data(), 0, n1 * 4, h.stream));
      HIP_TRY(hipMemsetAsync(p.far_mark.data(), 0, n1 * 4, h.stream));
      HIP_TRY(hipMemsetAsync(p.far_count.data(), 0, 2 * sizeof(uint32_t), h.stream));
      p.hi  = FLT_MAX;  // no window until the driver sets one (cugraph_amd/mg_traversal.py never does)
      p.win = 1;
      if (n_sources) hipLaunchKernelGGL(k_mg_sssp_sources, g, 256, 0, h.stream, source_rows, n_sources, p.st.data(), p.q_cur, p.cnt.data());
    }
    counters_t c{};
    h.read_back(&c, p.cnt.data(), 1);
    c.fold();
    p.n_frontier = c.n_next;
  });
}

// the launch half of expand: on completion `send` holds the candidates grouped by owner, totals[r] = (start of owner r, start of owner r + 1) and the
// counters the list's length -- all on the device.  The exported call below reads them back; the library's BFS driver hands them to the other ranks
// from the device instead (traversal_mg_driver.hip: no host in a top-down level's exchange)
namespace cga {
void mg_plan_expand_launch(cugraph_amd_traversal_mg_plan_t* plan)
{
    traversal_mg_plan& p = TP(plan);
    handle_t const& h    = *p.h;
    HIP_TRY(hipSetDevice(h.device));
    HIP_TRY(hipMemsetAsync(p.cnt.data(), 0, sizeof(counters_t), h.stream));
    int64_t const n = (int64_t)p.n_frontier;
    if (p.mode == 1) { ++p.round; p.n_local = 0; }
    if (n > 0) {
      int const g = expand_grid(h, n);
      timed_launch tl(h, p.mode == 0 ? "bfs_expand" : "sssp_relax");
      if (p.mode == 0) {
        mg_bfs_state s{p.seen.data(), p.touched.data(), p.cand_parent.data(), p.row_vertex, p.cand.data(), p.cnt.data(), p.with_pred ? 1 : 0};
        hipLaunchKernelGGL(k_mg_bfs_expand, g, TV_BLOCK, 0, h.stream, (int32_t const*)p.q_cur, n, p.offsets, p.indices, p.bigq.data(), s);
        hipLaunchKernelGGL(k_mg_bfs_expand_big, h.num_cus * 4, TV_BLOCK, 0, h.stream, (int32_t const*)p.bigq.data(), p.offsets, p.indices, s);
      } else {
        mg_sssp_state s{p.st.data(), p.cand_best.data(), p.touched.data(), p.weights, p.row_vertex, p.cand.data(), p.cnt.data(), p.cutoff,
                        (uint32_t)((size_t)p.rank * p.L), p.own_in_place ? (uint32_t)p.n_rows : 0u, p.mark.data(), p.q_next, p.round,
                        float_bits(p.hi), p.win, p.far_mark.data(), p.q_far.data(), p.far_count.data()};
        hipLaunchKernelGGL(k_mg_sssp_expand, g, TV_BLOCK, 0, h.stream, (int32_t const*)p.q_cur, n, p.offsets, p.indices, p.bigq.data(), s);
        hipLaunchKernelGGL(k_mg_sssp_expand_big, h.num_cus * 4, TV_BLOCK, 0, h.stream, (int32_t const*)p.bigq.data(), p.offsets, p.indices, s);
      }
    }
    // bucket by owner
    int const nb = MG_BUCKET_BLOCKS;
    hipLaunchKernelGGL(k_mg_bucket_count, nb, 256, 0, h.stream, (int32_t const*)p.cand.data(), (counters_t const*)p.cnt.data(), (uint32_t)p.L, p.P, p.block_hist.data());
    hipLaunchKernelGGL(k_mg_bucket_scan, 1, 1024, 0, h.stream, p.block_hist.data(), p.P, nb, p.totals.data());
    if (p.mode == 0)
      hipLaunchKernelGGL(k_mg_bucket_scatter<0>, nb, 256, 0, h.stream, (int32_t const*)p.cand.data(), (counters_t const*)p.cnt.data(), (uint32_t)p.L, p.P,
                         (uint32_t const*)p.block_hist.data(), p.cand_parent.data(), (unsigned long long*)nullptr, p.touched.data(), p.send);
    else
      hipLaunchKernelGGL(k_mg_bucket_scatter<1>, nb, 256, 0, h.stream, (int32_t const*)p.cand.data(), (counters_t const*)p.cnt.data(), (uint32_t)p.L, p.P,
                         (uint32_t const*)p.block_hist.data(), (int32_t*)nullptr, p.cand_best.data(), p.touched.data(), p.send);
}
unsigned long long const* mg_plan_totals(cugraph_amd_traversal_mg_plan_t* plan) { return TP(plan).totals.data(); }
size_t mg_plan_capacity(cugraph_amd_traversal_mg_plan_t* plan) { return TP(plan).capacity; }
}  // namespace cga

/* Expands the local frontier; on return send holds, grouped by owner rank, send_counts[r] tuples for rank r. */
extern "C" cugraph_error_code_t cugraph_amd_traversal_mg_plan_expand(cugraph_amd_traversal_mg_plan_t* plan, size_t* send_counts, cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(plan != nullptr && send_counts != nullptr, CUGRAPH_INVALID_INPUT, "NULL argument");
    mg_plan_expand_launch(plan);
    traversal_mg_plan& p = TP(plan);
    handle_t const& h    = *p.h;
    // counts: totals[r] = (start of r, start of r + 1); the list length closes the last bucket
    struct { unsigned long long t[MG_MAX_RANKS]; } tot;
    HIP_TRY(hipMemcpyAsync(h.pinned, p.totals.data(), sizeof(unsigned long long) * MG_MAX_RANKS, hipMemcpyDeviceToHost, h.stream));
    counters_t c{};
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(h.pinned) + 1024, p.cnt.data(), sizeof(counters_t), hipMemcpyDeviceToHost, h.stream));
    h.sync();
    std::memcpy(&tot, h.pinned, sizeof(tot));
    std::memcpy(&c, static_cast<char*>(h.pinned) + 1024, sizeof(c));
    c.fold();
    CGA_EXPECTS((size_t)c.n_next <= p.capacity, CUGRAPH_UNKNOWN_ERROR, "candidate list overflowed the send capacity");
    if (p.mode == 1) {
      p.n_local = c.n_far;  // owned destinations relaxed in place: the head of the next local frontier is written already
      CGA_EXPECTS(p.n_local <= p.n_rows, CUGRAPH_UNKNOWN_ERROR, "next local frontier overflowed");
    }
    for (int r = 0; r < p.P; ++r) {
      uint32_t const first = (uint32_t)tot.t[r];
      uint32_t const next  = r + 1 < p.P ? (uint32_t)(tot.t[r] >> 32) : c.n_next;
      send_counts[r]       = (size_t)(next - first);
    }
  });
}

// ---- the launch half and the host half of apply / bottom_up.  The exported calls below run one after the other with a read-back in
// between; the library's own BFS driver (traversal_mg_driver.hip) ships the level's counters to every rank FROM THE DEVICE between the two
// halves (mg_plan_counters) and adopts what came back, so a level costs one host synchronisation less.
namespace cga {
// (n_tuples_dev != nullptr: recv holds P slots of L tuples, slot s filled by sender s with n_tuples_dev[s] tuples -- counts no host has seen)
void mg_plan_apply_launch(cugraph_amd_traversal_mg_plan_t* plan, int32_t const* recv, size_t n_tuples, uint32_t level, uint32_t const* n_tuples_dev)
{
  traversal_mg_plan& p = TP(plan);
  CGA_EXPECTS(n_tuples_dev == nullptr || p.mode == 0, CUGRAPH_INVALID_INPUT, "device-side tuple counts are a BFS level's");
  handle_t const& h    = *p.h;
  HIP_TRY(hipSetDevice(h.device));
  if (p.mode == 1 && p.n_local > 0) {  // apply appends behind what expand put there
    counters_t z{};
    z.n_next = (uint32_t)p.n_local;
    std::memcpy(h.pinned, &z, sizeof(z));
    HIP_TRY(hipMemcpyAsync(p.cnt.data(), h.pinned, sizeof(z), hipMemcpyHostToDevice, h.stream));
  } else {
    HIP_TRY(hipMemsetAsync(p.cnt.data(), 0, sizeof(counters_t), h.stream));
  }
  if (p.mode == 0) HIP_TRY(hipMemsetAsync(p.newfront.data(), 0, p.L / 8, h.stream));
  if (n_tuples || n_tuples_dev) {
    int const g = n_tuples_dev ? std::max(h.num_cus * 8 / p.P, 8) : (int)((n_tuples + 255) / 256);
    if (p.mode == 0)
      hipLaunchKernelGGL(k_mg_bfs_apply, dim3((unsigned)g, n_tuples_dev ? (unsigned)p.P : 1u), dim3(256), 0, h.stream, recv, n_tuples, n_tuples_dev, (size_t)p.L, (int32_t)level, p.dist.data(), p.with_pred ? p.pred.data() : (int32_t*)nullptr,
                         p.q_next, p.newfront.data(), p.cnt.data(), p.offsets, p.in_offsets);
    else
      hipLaunchKernelGGL(k_mg_sssp_apply, g, 256, 0, h.stream, recv, n_tuples, p.round, p.st.data(), p.mark.data(), p.q_next, p.cnt.data(), float_bits(p.hi), p.win,
                         p.far_mark.data(), p.q_far.data(), p.far_count.data());
  }
}
void mg_plan_bottom_up_launch(cugraph_amd_traversal_mg_plan_t* plan, uint32_t const* front, uint32_t level)
{
  traversal_mg_plan& p = TP(plan);
  CGA_EXPECTS(p.mode == 0 && p.in_offsets != nullptr, CUGRAPH_INVALID_INPUT, "bottom-up levels were not enabled (cugraph_amd_traversal_mg_plan_set_bottom_up)");
  handle_t const& h = *p.h;
  HIP_TRY(hipSetDevice(h.device));
  HIP_TRY(hipMemsetAsync(p.cnt.data(), 0, sizeof(counters_t), h.stream));
  HIP_TRY(hipMemsetAsync(p.newfront.data(), 0, p.L / 8, h.stream));
  int64_t const nv = (int64_t)p.n_rows;
  if (nv > 0) {
    int const grid = (int)std::max<int64_t>(1, std::min<int64_t>(((nv + 63) / 64 + TV_WAVES * BU_GROUPS - 1) / (TV_WAVES * BU_GROUPS), (int64_t)h.num_cus * 16));
    timed_launch tl(h, "bfs_bottom_up");
    hipLaunchKernelGGL(k_bfs_bottom_up<false>, grid, TV_BLOCK, 0, h.stream, p.in_offsets, p.in_indices, p.offsets, nv, p.seen.data() + (size_t)p.rank * (p.L / 32),
                       front, p.newfront.data(), p.dist.data(), p.with_pred ? p.pred.data() : (int32_t*)nullptr, (int32_t)level, p.cnt.data(),
                       (unsigned long long*)nullptr, p.ext_of_g);
    // the next level may run top-down: its queue
    hipLaunchKernelGGL(k_bfs_bitmap_to_queue, grid_for((int64_t)(p.L / 32), TV_BLOCK, 512), TV_BLOCK, 0, h.stream, (uint32_t const*)p.newfront.data(), (int64_t)(p.L / 32),
                       p.q_next, p.cnt.data());
  }
}
// the level's counters on the device (counters_t: cursors + replica lines, not folded)
void const* mg_plan_counters(cugraph_amd_traversal_mg_plan_t* plan) { return TP(plan).cnt.data(); }
// the host half: the next local frontier is what the launch half queued (n_next rows), with these degree sums
void mg_plan_adopt_level(cugraph_amd_traversal_mg_plan_t* plan, size_t n_next, unsigned long long out_edges, unsigned long long in_edges)
{
  traversal_mg_plan& p = TP(plan);
  std::swap(p.q_cur, p.q_next);
  p.n_frontier = n_next;
  p.n_local    = 0;
  p.last_out   = out_edges;
  p.last_in    = in_edges;
}
}  // namespace cga

/* Folds n_tuples received candidates into the local rows; n_next = size of the next local frontier. */
extern "C" cugraph_error_code_t cugraph_amd_traversal_mg_plan_apply(cugraph_amd_traversal_mg_plan_t* plan, const int32_t* recv, size_t n_tuples,
                                                                    uint32_t level, size_t* n_next, cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(plan != nullptr && n_next != nullptr, CUGRAPH_INVALID_INPUT, "NULL argument");
    mg_plan_apply_launch(plan, recv, n_tuples, level, nullptr);
    traversal_mg_plan& p = TP(plan);
    counters_t c{};
    p.h->read_back(&c, p.cnt.data(), 1);
    c.fold();
    mg_plan_adopt_level(plan, c.n_next, c.out_edges, c.in_edges);
    *n_next = c.n_next;
  });
}

/* SSSP near / far (sssp_impl.cuh:376-561 partitioned): rows whose distance drops to a value at or beyond `hi` wait in a per-rank far pile instead
   of joining the next frontier.  The caller -- all ranks together -- runs rounds until the frontier is empty everywhere, asks every rank for
   (entries still beyond the window, their smallest distance), picks the next window from the global minimum and calls advance, which splits
   the pile: the rows below the new bound become the frontier.  The fixed point and the parents do not depend on the windows. */
extern "C" cugraph_error_code_t cugraph_amd_traversal_mg_plan_sssp_set_window(cugraph_amd_traversal_mg_plan_t* plan, double hi, cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(plan != nullptr && TP(plan).mode == 1, CUGRAPH_INVALID_INPUT, "SSSP plan expected");
    TP(plan).hi = hi >= (double)FLT_MAX ? FLT_MAX : (float)hi;
  });
}
extern "C" cugraph_error_code_t cugraph_amd_traversal_mg_plan_sssp_far_stats(cugraph_amd_traversal_mg_plan_t* plan, size_t* n_far, double* min_distance,
                                                                             cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(plan != nullptr && n_far != nullptr && min_distance != nullptr && TP(plan).mode == 1, CUGRAPH_INVALID_INPUT, "SSSP plan / NULL argument");
    traversal_mg_plan& p = TP(plan);
    handle_t const& h    = *p.h;
    HIP_TRY(hipSetDevice(h.device));
    uint32_t n = 0;
    h.read_back(&n, p.far_count.data(), 1);
    CGA_EXPECTS(n <= p.n_rows, CUGRAPH_UNKNOWN_ERROR, "far pile overflowed");
    dvec<uint32_t> out(2);
    uint32_t const init[2] = {0u, 0xFFFFFFFFu};
    std::memcpy(h.pinned, init, sizeof(init));
    HIP_TRY(hipMemcpyAsync(out.data(), h.pinned, sizeof(init), hipMemcpyHostToDevice, h.stream));
    if (n) hipLaunchKernelGGL(k_mg_sssp_far_min, (int)((n + 255) / 256), 256, 0, h.stream, (int32_t const*)p.q_far.data(), n, (unsigned long long const*)p.st.data(), float_bits(p.hi), out.data());
    uint32_t got[2];
    h.read_back(got, out.data(), 2);
    *n_far = got[0];
    float f;
    std::memcpy(&f, &got[1], 4);
    *min_distance = got[0] ? (double)f : (double)FLT_MAX;
  });
}
extern "C" cugraph_error_code_t cugraph_amd_traversal_mg_plan_sssp_advance(cugraph_amd_traversal_mg_plan_t* plan, double hi_new, size_t* n_frontier,
                                                                           cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(plan != nullptr && n_frontier != nullptr && TP(plan).mode == 1, CUGRAPH_INVALID_INPUT, "SSSP plan / NULL argument");
    traversal_mg_plan& p = TP(plan);
    handle_t const& h    = *p.h;
    HIP_TRY(hipSetDevice(h.device));
    CGA_EXPECTS(p.n_frontier == 0, CUGRAPH_INVALID_INPUT, "the window may move only when the local frontier is empty");
    uint32_t n = 0;
    h.read_back(&n, p.far_count.data(), 1);
    float const hi_old = p.hi;
    p.hi               = hi_new >= (double)FLT_MAX ? FLT_MAX : (float)hi_new;
    ++p.win;
    ++p.round;
    HIP_TRY(hipMemsetAsync(p.far_count.data(), 0, 2 * sizeof(uint32_t), h.stream));  // [0]: the new pile, [1]: the new frontier
    if (n)
      hipLaunchKernelGGL(k_mg_sssp_split, (int)((n + 255) / 256), 256, 0, h.stream, (int32_t const*)p.q_far.data(), n, (unsigned long long const*)p.st.data(), float_bits(hi_old),
                         float_bits(p.hi), p.round, p.win, p.mark.data(), p.far_mark.data(), p.q_cur, p.far_count.data() + 1, p.q_far2.data(), p.far_count.data());
    uint32_t got[2];
    h.read_back(got, p.far_count.data(), 2);
    std::swap(p.q_far, p.q_far2);
    HIP_TRY(hipMemsetAsync(p.far_count.data() + 1, 0, sizeof(uint32_t), h.stream));
    p.n_frontier = got[1];
    *n_frontier  = got[1];
  });
}

/* BFS: enables bottom-up levels.  in_offsets [n_rows + 1] / in_indices: the in-edges of the local rows, neighbours as compact global ids in
   ascending order of their EXTERNAL id (the first frontier member of a row is then the minimum-external-id parent -- the rule of the
   top-down levels), in_indices over-allocated by at least 8 entries; ext_of_g [comm_size * rows_per_rank]: external id of a compact id.
   The arrays stay owned by the caller.  Replaces the bottom-up branch of bfs_impl.cuh:587-805 for the partitioned case. */
extern "C" cugraph_error_code_t cugraph_amd_traversal_mg_plan_set_bottom_up(cugraph_amd_traversal_mg_plan_t* plan, const int32_t* in_offsets,
                                                                            const int32_t* in_indices, const int32_t* ext_of_g, cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(plan != nullptr && TP(plan).mode == 0, CUGRAPH_INVALID_INPUT, "BFS plan expected");
    CGA_EXPECTS(in_offsets != nullptr && in_indices != nullptr && ext_of_g != nullptr, CUGRAPH_INVALID_INPUT, "NULL argument");
    traversal_mg_plan& p = TP(plan);
    p.in_offsets = in_offsets;
    p.in_indices = in_indices;
    p.ext_of_g   = ext_of_g;
  });
}

/* BFS, one bottom-up level: every unvisited local row scans its in-neighbours against `front` (the all-gathered frontier bitmap,
   comm_size * rows_per_rank bits) and stops at the first member.  No candidate exchange: the discoveries are local; their bits are
   shared by the same all-gather as after a top-down level (frontier_bits / merge_visited).  n_found = size of the next local frontier. */
extern "C" cugraph_error_code_t cugraph_amd_traversal_mg_plan_bottom_up(cugraph_amd_traversal_mg_plan_t* plan, const uint32_t* front, uint32_t level,
                                                                        size_t* n_found, cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(plan != nullptr && front != nullptr && n_found != nullptr, CUGRAPH_INVALID_INPUT, "NULL argument");
    mg_plan_bottom_up_launch(plan, front, level);
    traversal_mg_plan& p = TP(plan);
    counters_t c{};
    p.h->read_back(&c, p.cnt.data(), 1);
    c.fold();  // (the kernel counts its discoveries in the replica lines: folded into n_next)
    CGA_EXPECTS(c.n_next == c.n_big, CUGRAPH_UNKNOWN_ERROR, "bottom-up level: discoveries and queue length differ");
    mg_plan_adopt_level(plan, c.n_next, c.out_edges, c.in_edges);
    *n_found = c.n_next;
  });
}

/* BFS: sums of the out- and in-degrees of the vertices the last apply / bottom_up call discovered on this rank (zero unless bottom-up
   levels are enabled) */
extern "C" cugraph_error_code_t cugraph_amd_traversal_mg_plan_last_degree_sums(cugraph_amd_traversal_mg_plan_t* plan, unsigned long long* out_edges,
                                                                               unsigned long long* in_edges, cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(plan != nullptr && out_edges != nullptr && in_edges != nullptr, CUGRAPH_INVALID_INPUT, "NULL argument");
    *out_edges = TP(plan).last_out;
    *in_edges  = TP(plan).last_in;
  });
}

/* BFS: device pointer to this rank's new-frontier bits of the last apply / reset (rows_per_rank / 32 words). */
extern "C" cugraph_error_code_t cugraph_amd_traversal_mg_plan_frontier_bits(cugraph_amd_traversal_mg_plan_t* plan, const uint32_t** bits,
                                                                            cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(plan != nullptr && bits != nullptr && TP(plan).mode == 0, CUGRAPH_INVALID_INPUT, "BFS plan expected");
    *bits = TP(plan).newfront.data();
  });
}

/* BFS: visited |= the all-gathered new-frontier bits of every rank (comm_size * rows_per_rank / 32 words). */
extern "C" cugraph_error_code_t cugraph_amd_traversal_mg_plan_merge_visited(cugraph_amd_traversal_mg_plan_t* plan, const uint32_t* gathered,
                                                                            cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(plan != nullptr && gathered != nullptr && TP(plan).mode == 0, CUGRAPH_INVALID_INPUT, "BFS plan expected");
    traversal_mg_plan& p = TP(plan);
    handle_t const& h    = *p.h;
    HIP_TRY(hipSetDevice(h.device));
    size_t const n = p.L * (size_t)p.P / 32;
    hipLaunchKernelGGL(k_mg_or, (int)((n + 255) / 256), 256, 0, h.stream, p.seen.data(), gathered, n);
    if (!p.caller_keeps_buffers) h.sync();  // (`gathered` may be the caller's temporary)
  });
}

/* distances (BFS: int32, INT32_MAX unreached; SSSP: float, FLT_MAX unreached) and predecessors (external ids, -1) of the local rows */
extern "C" cugraph_error_code_t cugraph_amd_traversal_mg_plan_results(cugraph_amd_traversal_mg_plan_t* plan, void* distances, int32_t* predecessors,
                                                                      cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(plan != nullptr && distances != nullptr, CUGRAPH_INVALID_INPUT, "NULL argument");
    traversal_mg_plan& p = TP(plan);
    handle_t const& h    = *p.h;
    HIP_TRY(hipSetDevice(h.device));
    if (p.n_rows) {
      int const g = (int)((p.n_rows + 255) / 256);
      if (p.mode == 0)
        hipLaunchKernelGGL(k_mg_bfs_results, g, 256, 0, h.stream, (int32_t const*)p.dist.data(), p.with_pred ? (int32_t const*)p.pred.data() : (int32_t const*)nullptr,
                           p.n_rows, static_cast<int32_t*>(distances), predecessors);
      else
        hipLaunchKernelGGL(k_mg_sssp_results, g, 256, 0, h.stream, (unsigned long long const*)p.st.data(), p.n_rows, static_cast<float*>(distances),
                           p.with_pred ? predecessors : (int32_t*)nullptr);
    }
    h.sync();
  });
}
