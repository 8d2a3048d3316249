/*
 * oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's algorithms for the
 * PageRank / BFS / SSSP hot path.  Nothing in the product path (cugraph_amd/, include/) may link,
 * import or call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Parity status: PINNED.  Checked (tests/test_oracle.py) against
 *   - the reference's C-API golden vectors (cpp/tests/c_api/pagerank_test.c:385-544,
 *     bfs_test.c:160-210, sssp_test.c:167-223) and pylibcugraph goldens
 *     (python/pylibcugraph/pylibcugraph/tests/test_pagerank.py:14-147), committed under tests/golden/;
 *   - the reference's own CPU reference functions compiled in place into oracle/_ref/libref.so
 *     (recipe oracle/build_ref.sh): pagerank_reference (cpp/tests/link_analysis/pagerank_test.cpp:33-121),
 *     bfs_reference (cpp/tests/traversal/bfs_test.cpp:32-70), sssp_reference
 *     (cpp/tests/traversal/sssp_test.cpp:33-75).
 *
 * Each function cites the reference file:line it follows.
 */
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------------
 * RMAT edge generator.  Algorithm = cpp/src/generators/generate_rmat_edgelist.cuh:85-103 (per edge,
 * per bit from the MSB: r0 > a+b sets the src bit; r1 > (srcbit ? c/(1-a-b) : a/(a+b)) sets the dst
 * bit; no clip-and-flip, no scramble).  The reference draws r0,r1 from raft::random (not vendored,
 * stream not reproducible here), so the RNG is ours: counter-based splitmix64 on (seed, edge, bit),
 * compared as 32-bit integers against integer thresholds so CPU and GPU agree bit-for-bit.
 * The HIP generator (cugraph_amd/csrc/rmat.hip) implements the same function.
 * ---------------------------------------------------------------------------------------------- */
static inline uint64_t splitmix64_at(uint64_t seed, uint64_t counter)
{
  uint64_t z = seed + (counter + 1) * 0x9E3779B97F4A7C15ull;
  z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z          = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static inline uint32_t prob_to_u32(double p)
{
  if (p <= 0.0) return 0u;
  if (p >= 1.0) return 0xFFFFFFFFu;
  return (uint32_t)(p * 4294967296.0);
}

void orc_rmat_thresholds(double a, double b, double c, uint32_t* t_ab, uint32_t* t_a_norm, uint32_t* t_c_norm)
{
  double a_plus_b = a + b;
  double a_norm   = a_plus_b > 0.0 ? a / a_plus_b : 0.0;
  double c_norm   = (1.0 - a_plus_b) > 0.0 ? c / (1.0 - a_plus_b) : 0.0;
  *t_ab           = prob_to_u32(a_plus_b);
  *t_a_norm       = prob_to_u32(a_norm);
  *t_c_norm       = prob_to_u32(c_norm);
}

void orc_rmat(int scale, uint64_t first_edge, uint64_t num_edges, double a, double b, double c,
              uint64_t seed, int32_t* src, int32_t* dst)
{
  uint32_t t_ab, t_an, t_cn;
  orc_rmat_thresholds(a, b, c, &t_ab, &t_an, &t_cn);
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < (int64_t)num_edges; ++k) {
    uint64_t i = first_edge + (uint64_t)k;
    int32_t s = 0, d = 0;
    for (int bit = scale - 1; bit >= 0; --bit) {
      uint64_t z  = splitmix64_at(seed, i * 64ull + (uint64_t)bit);
      uint32_t r0 = (uint32_t)(z >> 32), r1 = (uint32_t)z;
      int sb      = r0 > t_ab;
      int db      = r1 > (sb ? t_cn : t_an);
      s |= sb << bit;
      d |= db << bit;
    }
    src[k] = s;
    dst[k] = d;
  }
}

/* ------------------------------------------------------------------------------------------------
 * COO -> compressed sparse (major-sorted, neighbours ascending, multi-edges / self loops kept):
 * what sort_and_compress_edgelist produces (cpp/src/structure/detail/structure_utils.cuh:197-464,
 * pair sort at :435-447).  Ties between equal (major,minor) pairs keep input order (stable), which
 * only matters for the weight column of multi-edges.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int32_t minor; float w; } pair_f;
typedef struct { int32_t minor; double w; } pair_d;
static int cmp_pair_f(const void* x, const void* y) { int32_t a = ((const pair_f*)x)->minor, b = ((const pair_f*)y)->minor; return (a > b) - (a < b); }
static int cmp_i32(const void* x, const void* y) { int32_t a = *(const int32_t*)x, b = *(const int32_t*)y; return (a > b) - (a < b); }

/* stable insertion/merge: qsort is not stable, so sort (minor, original position) pairs */
typedef struct { int32_t minor; int32_t pos; } mp_t;
static int cmp_mp(const void* x, const void* y)
{
  const mp_t *a = (const mp_t*)x, *b = (const mp_t*)y;
  if (a->minor != b->minor) return (a->minor > b->minor) - (a->minor < b->minor);
  return (a->pos > b->pos) - (a->pos < b->pos);
}

/* weights may be NULL; wsize = 4 (float) or 8 (double) */
void orc_coo_to_cs(int64_t nv, int64_t ne, const int32_t* major, const int32_t* minor,
                   const void* weights, int wsize, int64_t* offsets, int32_t* indices, void* out_w)
{
  memset(offsets, 0, sizeof(int64_t) * (size_t)(nv + 1));
  for (int64_t e = 0; e < ne; ++e) offsets[major[e] + 1]++;
  for (int64_t v = 0; v < nv; ++v) offsets[v + 1] += offsets[v];
  int64_t* cur = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nv > 0 ? nv : 1));
  memcpy(cur, offsets, sizeof(int64_t) * (size_t)nv);
  int64_t* epos = (int64_t*)malloc(sizeof(int64_t) * (size_t)(ne > 0 ? ne : 1));
  for (int64_t e = 0; e < ne; ++e) { int64_t p = cur[major[e]]++; indices[p] = minor[e]; epos[p] = e; }
  free(cur);
#pragma omp parallel
  {
    mp_t* buf = NULL; size_t cap = 0;
    int64_t* tmp = NULL;
#pragma omp for schedule(dynamic, 1024)
    for (int64_t v = 0; v < nv; ++v) {
      int64_t lo = offsets[v], n = offsets[v + 1] - lo;
      if (n < 2) continue;
      if ((size_t)n > cap) { cap = (size_t)n * 2; buf = (mp_t*)realloc(buf, cap * sizeof(mp_t)); tmp = (int64_t*)realloc(tmp, cap * sizeof(int64_t)); }
      for (int64_t k = 0; k < n; ++k) { buf[k].minor = indices[lo + k]; buf[k].pos = (int32_t)k; }
      qsort(buf, (size_t)n, sizeof(mp_t), cmp_mp);
      for (int64_t k = 0; k < n; ++k) tmp[k] = epos[lo + buf[k].pos];
      for (int64_t k = 0; k < n; ++k) { indices[lo + k] = buf[k].minor; epos[lo + k] = tmp[k]; }
    }
    free(buf); free(tmp);
  }
  if (weights && out_w) {
    if (wsize == 4) { const float* w = (const float*)weights; float* o = (float*)out_w; for (int64_t p = 0; p < ne; ++p) o[p] = w[epos[p]]; }
    else            { const double* w = (const double*)weights; double* o = (double*)out_w; for (int64_t p = 0; p < ne; ++p) o[p] = w[epos[p]]; }
  }
  free(epos);
  (void)cmp_pair_f; (void)cmp_i32;
}

/* ------------------------------------------------------------------------------------------------
 * PageRank.  Loop order of detail::pagerank, cpp/src/link_analysis/pagerank_impl.cuh:224-327:
 *   old <- pr; dangling <- sum_{outw==0} pr; pr <- pr/(outw==0?1:outw);
 *   pr[dst] <- unvarying + sum_in src_val*[w*]alpha (:261-287, e_op :268-283 applies alpha per edge);
 *   personalization scatter-add (:289-309); diff <- sum |pr-old| (:311-318); ++iter;
 *   if diff<eps break; else if iter>=max_iter break (:320-326); converged = iter<max_iter (:329).
 * Initial vector 1/V (:422-426); a supplied initial guess is NOT renormalised by the tuple overload
 * the C API uses (:427-432).  out-weight sums: :180-198 (degree cast to weight_t when unweighted).
 * Same maths as the test reference pagerank_reference (cpp/tests/link_analysis/pagerank_test.cpp:33-121),
 * which divides w/out_w per edge (:97) and normalises the initial guess (:48-56).
 *
 * The graph is CSC (row = destination, indices = sources).  acc64 != 0 accumulates every sum in
 * double (the "truth" the fp32 GPU result is compared with); acc64 == 0 accumulates in WT
 * sequentially, bit-faithful to the reference's CPU test function.
 * ---------------------------------------------------------------------------------------------- */
#define ORC_PAGERANK(NAME, WT)                                                                        \
  int NAME(int64_t nv, const int64_t* offsets, const int32_t* indices, const WT* weights,            \
           const WT* precomputed_outw, int64_t n_pers, const int32_t* pers_v, const WT* pers_val,     \
           int has_initial_guess, double alpha_d, double epsilon_d, int64_t max_iter, int acc64,      \
           WT* pr, int64_t* iters_out)                                                                \
  {                                                                                                   \
    if (nv == 0) { *iters_out = 0; return 1; }                                                        \
    const WT alpha = (WT)alpha_d, epsilon = (WT)epsilon_d;                                            \
    if (!has_initial_guess)                                                                           \
      for (int64_t i = 0; i < nv; ++i) pr[i] = (WT)1.0 / (WT)nv;                                      \
    WT* outw = (WT*)calloc((size_t)nv, sizeof(WT));                                                   \
    if (precomputed_outw) memcpy(outw, precomputed_outw, sizeof(WT) * (size_t)nv);                    \
    else if (weights) {                                                                               \
      double* t = (double*)calloc((size_t)nv, sizeof(double));                                        \
      for (int64_t e = 0; e < offsets[nv]; ++e) { if (acc64) t[indices[e]] += (double)weights[e]; else outw[indices[e]] += weights[e]; } \
      if (acc64) for (int64_t i = 0; i < nv; ++i) outw[i] = (WT)t[i];                                 \
      free(t);                                                                                        \
    } else {                                                                                          \
      int64_t* deg = (int64_t*)calloc((size_t)nv, sizeof(int64_t));                                   \
      for (int64_t e = 0; e < offsets[nv]; ++e) deg[indices[e]]++;                                    \
      for (int64_t i = 0; i < nv; ++i) outw[i] = (WT)deg[i];                                          \
      free(deg);                                                                                      \
    }                                                                                                 \
    WT pers_sum = 0;                                                                                  \
    if (n_pers > 0) { double s = 0; WT sf = 0; for (int64_t i = 0; i < n_pers; ++i) { s += (double)pers_val[i]; sf += pers_val[i]; } pers_sum = acc64 ? (WT)s : sf; } \
    WT* old = (WT*)malloc(sizeof(WT) * (size_t)nv);                                                   \
    WT* x   = (WT*)malloc(sizeof(WT) * (size_t)nv);                                                   \
    int64_t iter = 0;                                                                                 \
    for (;;) {                                                                                        \
      memcpy(old, pr, sizeof(WT) * (size_t)nv);                                                       \
      WT dangling;                                                                                    \
      { double s = 0; WT sf = 0;                                                                      \
        for (int64_t i = 0; i < nv; ++i) if (outw[i] == (WT)0) { s += (double)pr[i]; sf += pr[i]; }   \
        dangling = acc64 ? (WT)s : sf; }                                                              \
      for (int64_t i = 0; i < nv; ++i) x[i] = pr[i] / (outw[i] == (WT)0 ? (WT)1 : outw[i]);           \
      WT unvarying = n_pers == 0 ? (dangling * alpha + (WT)(1.0 - alpha_d)) / (WT)nv : (WT)0;         \
      _Pragma("omp parallel for schedule(dynamic, 4096)")                                             \
      for (int64_t i = 0; i < nv; ++i) {                                                              \
        if (acc64) {                                                                                  \
          double s = 0;                                                                               \
          if (weights) for (int64_t j = offsets[i]; j < offsets[i + 1]; ++j) s += (double)(x[indices[j]] * weights[j] * alpha); \
          else         for (int64_t j = offsets[i]; j < offsets[i + 1]; ++j) s += (double)(x[indices[j]] * alpha);              \
          pr[i] = (WT)((double)unvarying + s);                                                        \
        } else {                                                                                      \
          WT s = unvarying;                                                                           \
          if (weights) for (int64_t j = offsets[i]; j < offsets[i + 1]; ++j) s += x[indices[j]] * weights[j] * alpha; \
          else         for (int64_t j = offsets[i]; j < offsets[i + 1]; ++j) s += x[indices[j]] * alpha;              \
          pr[i] = s;                                                                                  \
        }                                                                                             \
      }                                                                                               \
      for (int64_t i = 0; i < n_pers; ++i)                                                            \
        pr[pers_v[i]] += (dangling * alpha + (WT)(1.0 - alpha_d)) * (pers_val[i] / pers_sum);         \
      WT diff;                                                                                        \
      { double s = 0; WT sf = 0;                                                                      \
        for (int64_t i = 0; i < nv; ++i) { WT d = pr[i] - old[i]; d = d < 0 ? -d : d; s += (double)d; sf += d; } \
        diff = acc64 ? (WT)s : sf; }                                                                  \
      iter++;                                                                                         \
      if (diff < epsilon) break; else if (iter >= max_iter) break;                                    \
    }                                                                                                 \
    free(old); free(x); free(outw);                                                                   \
    *iters_out = iter;                                                                                \
    return iter < max_iter;                                                                           \
  }

ORC_PAGERANK(orc_pagerank_f32, float)
ORC_PAGERANK(orc_pagerank_f64, double)

/* ------------------------------------------------------------------------------------------------
 * BFS.  bfs_reference, cpp/tests/traversal/bfs_test.cpp:32-70, extended to several sources as
 * detail::bfs allows (cpp/src/traversal/bfs_impl.cuh:270-285: every source gets distance 0).
 * depth_limit is compared after incrementing (bfs_test.cpp:66, bfs_impl.cuh:867-868).
 * Unreached: distance INT32_MAX, predecessor -1.  Graph is CSR (row = source vertex).
 * Predecessors here are the first discoverer in frontier order, like the reference function; the
 * reference GPU path uses reduce_op::any (bfs_impl.cuh:467) so any valid parent is accepted by its
 * own tests (bfs_test.cpp:217-233).  orc_bfs_min_pred() gives the canonical minimum-id parent our
 * HIP path produces deterministically.
 * ---------------------------------------------------------------------------------------------- */
void orc_bfs(int64_t nv, const int64_t* offsets, const int32_t* indices, const int32_t* sources,
             int64_t n_sources, int64_t depth_limit, int32_t* dist, int32_t* pred)
{
  for (int64_t i = 0; i < nv; ++i) { dist[i] = INT32_MAX; pred[i] = -1; }
  int32_t* cur = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nv > 0 ? nv : 1));
  int32_t* nxt = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nv > 0 ? nv : 1));
  int64_t ncur = 0, nnxt = 0;
  for (int64_t i = 0; i < n_sources; ++i)
    if (dist[sources[i]] != 0) { dist[sources[i]] = 0; cur[ncur++] = sources[i]; }
  int64_t depth = 0;
  while (ncur > 0) {
    nnxt = 0;
    for (int64_t f = 0; f < ncur; ++f) {
      int32_t row = cur[f];
      for (int64_t j = offsets[row]; j < offsets[row + 1]; ++j) {
        int32_t nbr = indices[j];
        if (dist[nbr] == INT32_MAX) { dist[nbr] = (int32_t)(depth + 1); pred[nbr] = row; nxt[nnxt++] = nbr; }
      }
    }
    int32_t* t = cur; cur = nxt; nxt = t; ncur = nnxt;
    ++depth;
    if (depth >= depth_limit) break;
  }
  free(cur); free(nxt);
}

/* canonical predecessor: min u with dist[u]+1 == dist[v] and (u,v) an edge */
void orc_bfs_min_pred(int64_t nv, const int64_t* offsets, const int32_t* indices, const int32_t* dist, int32_t* pred)
{
  for (int64_t i = 0; i < nv; ++i) pred[i] = -1;
  for (int64_t u = 0; u < nv; ++u) {
    if (dist[u] == INT32_MAX) continue;
    for (int64_t j = offsets[u]; j < offsets[u + 1]; ++j) {
      int32_t v = indices[j];
      if (dist[v] != INT32_MAX && dist[v] == dist[u] + 1 && dist[v] != 0 && (pred[v] == -1 || (int32_t)u < pred[v])) pred[v] = (int32_t)u;
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * SSSP.  Dijkstra, sssp_reference, cpp/tests/traversal/sssp_test.cpp:33-75: strict relax
 * new < min(d[nbr], cutoff) (:61-63; GPU path sssp_impl.cuh:58-71), unreached = type max, pred -1.
 * Binary heap keyed on (distance, vertex) like std::priority_queue<tuple, greater>.
 * ---------------------------------------------------------------------------------------------- */
#define ORC_SSSP(NAME, WT, WMAX)                                                                      \
  void NAME(int64_t nv, const int64_t* offsets, const int32_t* indices, const WT* weights,            \
            int32_t source, double cutoff_d, WT* dist, int32_t* pred)                                 \
  {                                                                                                   \
    typedef struct { WT d; int32_t v; } item;                                                         \
    WT cutoff = cutoff_d >= (double)WMAX ? WMAX : (WT)cutoff_d;                                       \
    for (int64_t i = 0; i < nv; ++i) { dist[i] = WMAX; pred[i] = -1; }                                \
    size_t cap = 1024, n = 0;                                                                         \
    item* h = (item*)malloc(cap * sizeof(item));                                                      \
    dist[source] = (WT)0; h[n].d = (WT)0; h[n].v = source; n++;                                       \
    while (n > 0) {                                                                                   \
      item top = h[0]; item last = h[--n];                                                            \
      size_t i = 0;                                                                                   \
      for (;;) { size_t l = 2 * i + 1, r = l + 1, m = i; item mv = last;                              \
        if (l < n && (h[l].d < mv.d || (h[l].d == mv.d && h[l].v < mv.v))) { m = l; mv = h[l]; }      \
        if (r < n && (h[r].d < mv.d || (h[r].d == mv.d && h[r].v < mv.v))) { m = r; mv = h[r]; }      \
        if (m == i) break; h[i] = h[m]; i = m; }                                                      \
      if (n > 0) h[i] = last;                                                                         \
      if (top.d > dist[top.v]) continue;                                                              \
      for (int64_t j = offsets[top.v]; j < offsets[top.v + 1]; ++j) {                                 \
        int32_t nbr = indices[j]; WT nd = top.d + weights[j];                                         \
        WT thr = dist[nbr] < cutoff ? dist[nbr] : cutoff;                                             \
        if (nd < thr) {                                                                               \
          dist[nbr] = nd; pred[nbr] = top.v;                                                          \
          if (n == cap) { cap *= 2; h = (item*)realloc(h, cap * sizeof(item)); }                      \
          size_t k = n++; item it; it.d = nd; it.v = nbr;                                             \
          while (k > 0) { size_t p = (k - 1) / 2;                                                     \
            if (h[p].d < it.d || (h[p].d == it.d && h[p].v <= it.v)) break; h[k] = h[p]; k = p; }     \
          h[k] = it;                                                                                  \
        }                                                                                             \
      }                                                                                               \
    }                                                                                                 \
    free(h);                                                                                          \
  }

ORC_SSSP(orc_sssp_f32, float, FLT_MAX)
ORC_SSSP(orc_sssp_f64, double, DBL_MAX)

/* canonical predecessor: lexicographic minimum (d[u]+w, u) over in-edges, the reference's
 * reduce_op::minimum<tuple<distance,pred>> (sssp_impl.cuh:334); source keeps -1. */
#define ORC_SSSP_MIN_PRED(NAME, WT, WMAX)                                                             \
  void NAME(int64_t nv, const int64_t* offsets, const int32_t* indices, const WT* weights,            \
            int32_t source, const WT* dist, int32_t* pred)                                            \
  {                                                                                                   \
    for (int64_t i = 0; i < nv; ++i) pred[i] = -1;                                                    \
    for (int64_t u = 0; u < nv; ++u) {                                                                \
      if (dist[u] == WMAX) continue;                                                                  \
      for (int64_t j = offsets[u]; j < offsets[u + 1]; ++j) {                                         \
        int32_t v = indices[j];                                                                       \
        if (v == source || dist[v] == WMAX) continue;                                                 \
        if (dist[u] + weights[j] == dist[v] && (pred[v] == -1 || (int32_t)u < pred[v])) pred[v] = (int32_t)u; \
      }                                                                                               \
    }                                                                                                 \
  }
ORC_SSSP_MIN_PRED(orc_sssp_min_pred_f32, float, FLT_MAX)
ORC_SSSP_MIN_PRED(orc_sssp_min_pred_f64, double, DBL_MAX)

/* ------------------------------------------------------------------------------------------------
 * Louvain, C twin of oracle.py: _louvain_level / louvain for graphs too large for the numpy version.
 * Follows detail::louvain (cpp/src/community/louvain_impl.cuh:78-262) with rng_state = nullopt: synchronous local moving --
 * every vertex takes the neighbouring cluster with the largest modularity gain (update_clustering_by_delta_modularity,
 * detail/common_methods.cuh:259-447; gain expression :70-125), ties to the smaller cluster id, moves only "up" or only "down"
 * in alternate sweeps (a sweep with no move in its direction applies the other direction's; that flip is local to the sweep: up_down is
 * a by-value argument, :277) -- a modularity test per sweep (compute_modularity, :176-228) and contraction per level
 * (graph_contraction, :230-257), whose coarse vertices are numbered BY DEGREE, descending (coarsen_graph(renumber = true) ->
 * renumber_edgelist_impl.cuh step 4; equal degrees in label order).  Pinned to the reference's two C-API goldens AND its three karate
 * goldens (cpp/tests/community/louvain_test.cpp:228-237) by tests/test_oracle.py.  Per vertex the weights are accumulated per neighbouring cluster in STORED EDGE ORDER (a
 * marker array instead of the numpy version's lexsort: same sums in the same order), clusters are then visited in ascending
 * order.  Edges must be grouped by source (any order inside a source).
 * DEVIATION from the reference, on purpose: every sum and every gain here is fp64 whatever the graph's weight type; the reference computes
 * them in weight_t (float for FLOAT32 graphs: common_methods.cuh:71-97, its thrust / cub reduction order unpinned).  fp64 is the wider,
 * order-independent choice the library makes too (64-bit fixed point: exact for the integer-weight graphs the fixtures use), so the
 * restatement and the library agree to the bit; against the reference itself the parity claim rests on its own goldens
 * (cpp/tests/c_api/louvain_test.c, tests/test_oracle.py) -- on float graphs whose best moves are decided by less than float rounding the
 * reference may take other moves.
 * ---------------------------------------------------------------------------------------------- */
/* louvain_delta_modularity_noise_floor (common_methods.cuh:52-58): 1e-12 for float graphs, 1e-15 for double */
static double lv_noise_floor = 1e-15;
void orc_louvain_set_noise_floor(double f) { lv_noise_floor = f; }
typedef struct { int32_t c; double w; } lv_cw;
static int cmp_lv_cw(const void* x, const void* y) { int32_t a = ((const lv_cw*)x)->c, b = ((const lv_cw*)y)->c; return (a > b) - (a < b); }

static double lv_q(int64_t ne, const int32_t* src, const int32_t* dst, const double* w, const int32_t* c, int64_t nv, const double* a, double m, double res)
{
  double internal = 0.0, sq = 0.0;
  for (int64_t i = 0; i < ne; ++i) if (c[src[i]] == c[dst[i]]) internal += w[i];
  for (int64_t i = 0; i < nv; ++i) sq += a[i] * a[i];
  return internal / m - (res * sq) / (m * m);
}

/* one level: accepted clustering (labels = vertex ids of the level) and the modularity it reached */
static double lv_level(int64_t nv, int64_t ne, const int32_t* src, const int32_t* dst, const double* w, double threshold, double res, double m,
                       int32_t* accepted, int* sweeps)
{
  int64_t* off = (int64_t*)calloc((size_t)nv + 1, sizeof(int64_t));
  for (int64_t i = 0; i < ne; ++i) off[src[i] + 1]++;
  for (int64_t v = 0; v < nv; ++v) off[v + 1] += off[v];
  double* k = (double*)calloc((size_t)nv, sizeof(double));
  for (int64_t i = 0; i < ne; ++i) k[src[i]] += w[i];
  int32_t* c = (int32_t*)malloc(sizeof(int32_t) * (size_t)nv);
  int32_t* best_c = (int32_t*)malloc(sizeof(int32_t) * (size_t)nv);
  double* best_d = (double*)malloc(sizeof(double) * (size_t)nv);
  double* a = (double*)malloc(sizeof(double) * (size_t)nv);
  int64_t* mark = (int64_t*)malloc(sizeof(int64_t) * (size_t)nv);
  double* acc = (double*)malloc(sizeof(double) * (size_t)nv);
  int64_t maxdeg = 0;
  for (int64_t v = 0; v < nv; ++v) { c[v] = (int32_t)v; accepted[v] = (int32_t)v; a[v] = k[v]; mark[v] = -1; if (off[v + 1] - off[v] > maxdeg) maxdeg = off[v + 1] - off[v]; }
  lv_cw* lst = (lv_cw*)malloc(sizeof(lv_cw) * (size_t)(maxdeg > 0 ? maxdeg : 1));
  double new_q = lv_q(ne, src, dst, w, c, nv, a, m, res), cur_q = new_q - 1.0;
  int up_down = 1;
  double min_gain = threshold / (double)(nv > 0 ? nv : 1);
  if (min_gain < lv_noise_floor) min_gain = lv_noise_floor;
  while (new_q > cur_q + threshold) {
    cur_q = new_q;
    ++*sweeps;
    for (int64_t v = 0; v < nv; ++v) {
      int32_t cv = c[v];
      double old_sum = 0.0, sub = 0.0;
      int64_t n = 0;
      int64_t const stamp = (int64_t)(*sweeps) * nv + v;  /* unique per (sweep, vertex): marks of earlier sweeps must not match */
      for (int64_t p = off[v]; p < off[v + 1]; ++p) {
        int32_t u = dst[p], cl = c[u];
        if (mark[cl] != stamp) { mark[cl] = stamp; acc[cl] = 0.0; lst[n++].c = cl; }
        acc[cl] += w[p];
        if (u == (int32_t)v) sub += w[p];
        else if (cl == cv) old_sum += w[p];
      }
      qsort(lst, (size_t)n, sizeof(lv_cw), cmp_lv_cw);
      int32_t bc = -1;
      double bd = 0.0, kk = k[v], a_old = a[cv];
      for (int64_t j = 0; j < n; ++j) {
        int32_t cl = lst[j].c;
        double s = acc[cl];
        double new_sum = cl == cv ? s - sub : s;
        double delta = 2.0 * (((new_sum - old_sum) / m) - res * (a[cl] * kk - a_old * kk + kk * kk) / (m * m));
        if (delta > bd) { bd = delta; bc = cl; }
      }
      best_c[v] = bc; best_d[v] = bd;
    }
    int64_t moves = 0;
    for (int64_t v = 0; v < nv; ++v) moves += best_d[v] > min_gain && ((best_c[v] > c[v]) == (up_down != 0));
    /* update_clustering_by_delta_modularity takes up_down BY VALUE (common_methods.cuh:277, 445): a sweep without a move in its direction applies
     * the moves of the other one, and that flip does not outlive the sweep */
    int const dir = moves == 0 ? !up_down : up_down;
    for (int64_t v = 0; v < nv; ++v) if (best_d[v] > min_gain && ((best_c[v] > c[v]) == (dir != 0))) c[v] = best_c[v];
    for (int64_t v = 0; v < nv; ++v) a[v] = 0.0;
    for (int64_t v = 0; v < nv; ++v) a[c[v]] += k[v];
    up_down = !up_down;
    new_q = lv_q(ne, src, dst, w, c, nv, a, m, res);
    if (getenv("ORC_LOUVAIN_TRACE")) fprintf(stderr, "[orc louvain] sweep %d: moves %lld, Q %.17g -> %.17g\n", *sweeps, (long long)moves, cur_q, new_q);
    if (new_q > cur_q + threshold) memcpy(accepted, c, sizeof(int32_t) * (size_t)nv);
  }
  free(off); free(k); free(c); free(best_c); free(best_d); free(a); free(mark); free(acc); free(lst);
  return cur_q;
}

/* returns the number of levels; clusters[nv], *modularity, *total_sweeps */
int orc_louvain(int64_t nv, int64_t ne, const int32_t* src_in, const int32_t* dst_in, const double* w_in, int64_t max_level, double threshold,
                double resolution, int32_t* clusters, double* modularity, int* total_sweeps)
{
  size_t e1 = (size_t)(ne > 0 ? ne : 1);
  int32_t* src = (int32_t*)malloc(sizeof(int32_t) * e1);
  int32_t* dst = (int32_t*)malloc(sizeof(int32_t) * e1);
  double* w = (double*)malloc(sizeof(double) * e1);
  double m = 0.0;
  for (int64_t i = 0; i < ne; ++i) { src[i] = src_in[i]; dst[i] = dst_in[i]; w[i] = w_in ? w_in[i] : 1.0; }
  for (int64_t i = 0; i < ne; ++i) m += w[i];
  for (int64_t v = 0; v < nv; ++v) clusters[v] = (int32_t)v;
  double best = -1.0;
  int levels = 0;
  int64_t cur_nv = nv, cur_ne = ne;
  *total_sweeps = 0;
  while (levels < max_level && cur_nv > 0 && m > 0.0) {
    ++levels;
    int32_t* c = (int32_t*)malloc(sizeof(int32_t) * (size_t)cur_nv);
    double q = lv_level(cur_nv, cur_ne, src, dst, w, threshold, resolution, m, c, total_sweeps);
    if (q <= best) { free(c); break; }
    best = q;
    /* new id = rank of the label among the labels in use */
    int32_t* rank = (int32_t*)calloc((size_t)cur_nv + 1, sizeof(int32_t));
    for (int64_t v = 0; v < cur_nv; ++v) rank[c[v]] = 1;
    int32_t ncl = 0;
    for (int64_t v = 0; v < cur_nv; ++v) { int32_t u = rank[v]; rank[v] = ncl; ncl += u; }
    for (int64_t v = 0; v < cur_nv; ++v) c[v] = rank[c[v]];
    for (int64_t v = 0; v < nv; ++v) clusters[v] = c[clusters[v]];
    /* coarse edges: (cs, cd) ascending, weight = sum in stored order.  Stable counting sort by cs, then a marker pass per cs */
    int64_t* coff = (int64_t*)calloc((size_t)ncl + 1, sizeof(int64_t));
    for (int64_t i = 0; i < cur_ne; ++i) coff[c[src[i]] + 1]++;
    for (int32_t v = 0; v < ncl; ++v) coff[v + 1] += coff[v];
    int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * ((size_t)ncl + 1));
    memcpy(fill, coff, sizeof(int64_t) * ((size_t)ncl + 1));
    int32_t* gd = (int32_t*)malloc(sizeof(int32_t) * e1);
    double* gw = (double*)malloc(sizeof(double) * e1);
    for (int64_t i = 0; i < cur_ne; ++i) { int64_t p = fill[c[src[i]]]++; gd[p] = c[dst[i]]; gw[p] = w[i]; }
    int64_t* mark = (int64_t*)malloc(sizeof(int64_t) * (size_t)(ncl > 0 ? ncl : 1));
    double* acc = (double*)malloc(sizeof(double) * (size_t)(ncl > 0 ? ncl : 1));
    for (int32_t v = 0; v < ncl; ++v) mark[v] = -1;
    int64_t maxg = 0;
    for (int32_t v = 0; v < ncl; ++v) if (coff[v + 1] - coff[v] > maxg) maxg = coff[v + 1] - coff[v];
    lv_cw* lst = (lv_cw*)malloc(sizeof(lv_cw) * (size_t)(maxg > 0 ? maxg : 1));
    int64_t out = 0;
    for (int32_t cs = 0; cs < ncl; ++cs) {
      int64_t n = 0;
      for (int64_t p = coff[cs]; p < coff[cs + 1]; ++p) {
        int32_t cd = gd[p];
        if (mark[cd] != cs) { mark[cd] = cs; acc[cd] = 0.0; lst[n++].c = cd; }
        acc[cd] += gw[p];
      }
      qsort(lst, (size_t)n, sizeof(lv_cw), cmp_lv_cw);
      for (int64_t j = 0; j < n; ++j) { src[out] = cs; dst[out] = lst[j].c; w[out] = acc[lst[j].c]; ++out; }  /* out <= cur_ne: in place is safe (reads are in gd/gw) */
      coff[cs] = out - n;  /* (row cs of the fine list has been consumed: from here on coff[cs] = start of COARSE row cs) */
    }
    coff[ncl] = out;
    /* graph_contraction (common_methods.cuh:231-263) = coarsen_graph(..., renumber = true): the coarse vertices are numbered as every graph creation
     * numbers vertices -- by degree, descending, equal degrees in ascending label order (renumber_edgelist_impl.cuh, step 4: a stable key sort over
     * the id-sorted vertex list); degree = coarse edges leaving the vertex = its distinct neighbour clusters.  The ids decide the ties (smaller
     * cluster id wins) and the up / down rule of the next level: part of the algorithm (the reference's karate goldens at resolution 1,
     * cpp/tests/community/louvain_test.cpp:228-237, are missed with label-order ids). */
    {
      int64_t maxd = 0;
      for (int32_t v = 0; v < ncl; ++v) if (coff[v + 1] - coff[v] > maxd) maxd = coff[v + 1] - coff[v];
      int64_t* bin = (int64_t*)calloc((size_t)maxd + 2, sizeof(int64_t));  /* counting sort by (maxd - degree): stable, so labels stay ascending */
      for (int32_t v = 0; v < ncl; ++v) bin[maxd - (coff[v + 1] - coff[v]) + 1]++;
      for (int64_t d = 0; d <= maxd; ++d) bin[d + 1] += bin[d];
      int32_t* new_id = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ncl > 0 ? ncl : 1));
      int32_t* old_of = (int32_t*)malloc(sizeof(int32_t) * (size_t)(ncl > 0 ? ncl : 1));
      for (int32_t v = 0; v < ncl; ++v) { int64_t const at = bin[maxd - (coff[v + 1] - coff[v])]++; new_id[v] = (int32_t)at; old_of[at] = v; }
      for (int64_t v = 0; v < nv; ++v) clusters[v] = new_id[clusters[v]];
      /* rows in the new order, every row ascending in the new neighbour ids (gd / gw are free again: the coarse list is in src / dst / w) */
      int64_t at = 0;
      for (int32_t r = 0; r < ncl; ++r) {
        int32_t const o = old_of[r];
        int64_t const n = coff[o + 1] - coff[o];
        for (int64_t j = 0; j < n; ++j) { lst[j].c = new_id[dst[coff[o] + j]]; lst[j].w = w[coff[o] + j]; }
        qsort(lst, (size_t)n, sizeof(lv_cw), cmp_lv_cw);
        for (int64_t j = 0; j < n; ++j) { gd[at + j] = lst[j].c; gw[at + j] = lst[j].w; }
        at += n;
      }
      at = 0;
      for (int32_t r = 0; r < ncl; ++r) {
        int64_t const n = coff[old_of[r] + 1] - coff[old_of[r]];
        for (int64_t j = 0; j < n; ++j) { src[at + j] = r; dst[at + j] = gd[at + j]; w[at + j] = gw[at + j]; }
        at += n;
      }
      free(bin); free(new_id); free(old_of);
    }
    cur_ne = out;
    cur_nv = ncl;
    free(c); free(rank); free(coff); free(fill); free(gd); free(gw); free(mark); free(acc); free(lst);
  }
  *modularity = best;
  free(src); free(dst); free(w);
  return levels;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
