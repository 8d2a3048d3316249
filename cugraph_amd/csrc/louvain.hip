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
// of the level's edges by (source, cluster of destination) followed by sequential segment sums -- no hash maps, and the
// summation order is fixed, so the result is bit-reproducible and equal to the oracle's (oracle/oracle.py: louvain) whenever
// the two see the same edge order.  All arithmetic is fp64, expressions in the reference's operation order, contraction off.
// Cluster labels of a contracted level = rank of the old label among the labels in use (the reference's labels are whatever
// its coarsen_graph renumbering assigns; its two C-API goldens come out identically, labels included).
// First version: one thread walks one vertex's (sorted) edges, so a hub row is serial -- fine up to ~10^7 edges, not tuned.
#pragma clang fp contract(off)
#include "common.hpp"

#include <algorithm>
#include <cmath>
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
{
  LV_LOOP(v, nv) for (int32_t p = offsets[v]; p < offsets[v + 1]; ++p) src[p] = (int32_t)v;
}
template <typename WT>
__global__ void k_to_double(WT const* w, int64_t n, double* out) { LV_LOOP(i, n) out[i] = w ? (double)w[i] : 1.0; }

// keys (hi[i] or map_hi[hi[i]]) << 32 | (lo[i] or map_lo[lo[i]]), payload i
__global__ void k_pair_keys(int32_t const* hi, int32_t const* map_hi, int32_t const* lo, int32_t const* map_lo, int64_t n, uint64_t* keys, uint32_t* vals)
{
  LV_LOOP(i, n)
  {
    uint32_t const a = (uint32_t)(map_hi ? map_hi[hi[i]] : hi[i]), b = (uint32_t)(map_lo ? map_lo[lo[i]] : lo[i]);
    keys[i] = ((uint64_t)a << 32) | b;
    vals[i] = (uint32_t)i;
  }
}
__global__ void k_single_keys(int32_t const* c, int64_t n, uint64_t* keys, uint32_t* vals)
{
  LV_LOOP(i, n) { keys[i] = (uint32_t)c[i]; vals[i] = (uint32_t)i; }
}
__global__ void k_heads(uint64_t const* keys, int64_t n, uint32_t* head) { LV_LOOP(i, n) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u; }

// vertex weights: sequential sum of the vertex's edges in stored order (edges are grouped by source)
__global__ void k_vertex_weights(int32_t const* off, double const* w, int64_t nv, double* k)
{
  LV_LOOP(v, nv)
  {
    double s = 0.0;
    for (int32_t p = off[v]; p < off[v + 1]; ++p) s += w[p];
    k[v] = s;
  }
}

// cluster weights: vertices sorted by cluster (stable); the head of a segment sums its members' weights in vertex order
__global__ void k_cluster_weights(uint64_t const* keys, uint32_t const* perm, uint32_t const* head, double const* k, int64_t nv, double* a)
{
  LV_LOOP(i, nv)
  {
    if (!head[i]) continue;
    double s = 0.0;
    int64_t j = i;
    do { s += k[perm[j]]; ++j; } while (j < nv && !head[j]);
    a[(uint32_t)keys[i]] = s;
  }
}

// One thread per vertex over its edges sorted by (vertex, cluster of destination, stored order):
// detail::key_aggregated_edge_op_t + reduce_op_t (common_methods.cuh:70-125) after the old_cluster_sum / cluster_subtract pass (:335-362)
__global__ void k_best_move(int32_t const* off, uint64_t const* keys, uint32_t const* perm, int32_t const* dst, double const* w, int32_t const* c,
                            double const* k, double const* a, double m, double resolution, int64_t nv, int32_t* best_c, double* best_d)
{
  LV_LOOP(v, nv)
  {
    int32_t const b = off[v], e = off[v + 1];
    int32_t const cv = c[v];
    double old_sum = 0.0, sub = 0.0;
    for (int32_t p = b; p < e; ++p) {
      uint32_t const ep = perm[p];
      if (dst[ep] == (int32_t)v) sub += w[ep];
      else if ((int32_t)(uint32_t)keys[p] == cv) old_sum += w[ep];
    }
    double const kk = k[v], a_old = a[cv];
    int32_t bc = -1;
    double bd  = 0.0;
    int32_t p = b;
    while (p < e) {
      int32_t const cl = (int32_t)(uint32_t)keys[p];
      double s = 0.0;
      do { s += w[perm[p]]; ++p; } while (p < e && (int32_t)(uint32_t)keys[p] == cl);
      double const new_sum = cl == cv ? s - sub : s;
      double const a_new   = a[cl];
      double const delta   = 2.0 * (((new_sum - old_sum) / m) - resolution * (a_new * kk - a_old * kk + kk * kk) / (m * m));
      if (delta > bd) { bd = delta; bc = cl; }  // clusters ascend: ties keep the smaller id
    }
    best_c[v] = bc;
    best_d[v] = bd;
  }
}

__global__ void k_count_moves(int32_t const* c, int32_t const* best_c, double const* best_d, double min_gain, int up_down, int64_t nv, uint32_t* count)
{
  LV_LOOP(v, nv)
  {
    bool const move = best_d[v] > min_gain && ((best_c[v] > c[v]) == (up_down != 0));
    uint64_t const mm = __ballot(move);
    if ((threadIdx.x & 63) == (unsigned)(__ffsll((unsigned long long)mm) - 1) && mm) atomicAdd(count, (uint32_t)__popcll(mm));
  }
}
__global__ void k_apply_moves(int32_t* c, int32_t const* best_c, double const* best_d, double min_gain, int up_down, int64_t nv)
{
  LV_LOOP(v, nv) if (best_d[v] > min_gain && ((best_c[v] > c[v]) == (up_down != 0))) c[v] = best_c[v];
}

// fixed-order reductions (one workgroup): sum of w over intra-cluster edges; sum of squares
__global__ void __launch_bounds__(1024) k_sum_internal(int32_t const* src, int32_t const* dst, double const* w, int32_t const* c, int64_t ne, double* out)
{
  __shared__ double red[1024];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < ne; i += 1024) s += c[src[i]] == c[dst[i]] ? w[i] : 0.0;
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) *out = red[0];
}
__global__ void __launch_bounds__(1024) k_sum_squares(double const* a, int64_t n, double* out)
{
  __shared__ double red[1024];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) s += a[i] * a[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) *out = red[0];
}
__global__ void __launch_bounds__(1024) k_sum_all(double const* w, int64_t n, double* out)
{
  __shared__ double red[1024];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) s += w[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) *out = red[0];
}

__global__ void k_mark_labels(int32_t const* c, int64_t nv, uint32_t* used) { LV_LOOP(v, nv) used[c[v]] = 1u; }
__global__ void k_relabel(int32_t* c, uint32_t const* rank, int64_t nv) { LV_LOOP(v, nv) c[v] = (int32_t)rank[c[v]]; }
__global__ void k_compose(int32_t* part, int32_t const* c, int64_t n) { LV_LOOP(i, n) part[i] = c[part[i]]; }
__global__ void k_copy_i32(int32_t* dst, int32_t const* src, int64_t n) { LV_LOOP(i, n) dst[i] = src[i]; }

// contraction: edges sorted by (cluster of src, cluster of dst); the head of a segment emits one coarse edge whose weight is
// the sequential sum of the segment
__global__ void k_coarse_edges(uint64_t const* keys, uint32_t const* perm, uint32_t const* head, uint32_t const* pos, double const* w, int64_t ne,
                               int32_t* csrc, int32_t* cdst, double* cw)
{
  LV_LOOP(i, ne)
  {
    if (!head[i]) continue;
    double s = 0.0;
    int64_t j = i;
    do { s += w[perm[j]]; ++j; } while (j < ne && !head[j]);
    uint32_t const o = pos[i];
    csrc[o] = (int32_t)(keys[i] >> 32);
    cdst[o] = (int32_t)(uint32_t)keys[i];
    cw[o]   = s;
  }
}

struct level_t {
  int64_t nv{0}, ne{0};
  dvec<int32_t> src, dst, off;
  dvec<double> w;
};

void build_offsets(handle_t const& h, level_t& L)
{
  L.off.resize_discard((size_t)L.nv + 1);
  dvec<uint32_t> cnt((size_t)L.nv + 1);
  HIP_TRY(hipMemsetAsync(cnt.data(), 0, ((size_t)L.nv + 1) * sizeof(uint32_t), h.stream));
  if (L.ne > 0) histogram_i32(h, L.src.data(), L.ne, cnt.data());
  exclusive_scan_u32(h, cnt.data(), reinterpret_cast<uint32_t*>(L.off.data()), L.nv + 1);
}

void sort_pairs(handle_t const& h, dvec<uint64_t>& keys, dvec<uint32_t>& vals, int64_t n, int bits_hi, int bits_lo)
{
  if (n <= 1) return;
  dvec<uint64_t> kt((size_t)n);
  dvec<uint32_t> vt((size_t)n);
  radix_sort_u64_u32(h, keys.data(), vals.data(), kt.data(), vt.data(), n, 0, bits_lo);
  if (bits_hi > 0) radix_sort_u64_u32(h, keys.data(), vals.data(), kt.data(), vt.data(), n, 32, 32 + bits_hi);
}

double read_double(handle_t const& h, double const* p)
{
  double v = 0;
  h.read_back(&v, p, 1);
  return v;
}

// one level (the body of the while loop of detail::louvain, louvain_impl.cuh:78-262): accepted clustering and its modularity
double run_level(handle_t const& h, level_t const& L, double m, double threshold, double resolution, dvec<int32_t>& accepted)
{
  int64_t const nv = L.nv, ne = L.ne;
  int const g_v = grid_for(nv, kBlock, 8192), g_e = grid_for(ne, kBlock, 8192);
  dvec<double> k((size_t)nv), a((size_t)nv), best_d((size_t)nv), scal(2);
  dvec<int32_t> c((size_t)nv), best_c((size_t)nv);
  dvec<uint32_t> count(1), vhead((size_t)nv), vperm((size_t)nv), eperm((size_t)std::max<int64_t>(ne, 1));
  dvec<uint64_t> vkeys((size_t)nv), ekeys((size_t)std::max<int64_t>(ne, 1));
  accepted.resize_discard((size_t)nv);
  hipLaunchKernelGGL(k_vertex_weights, g_v, kBlock, 0, h.stream, (int32_t const*)L.off.data(), (double const*)L.w.data(), nv, k.data());
  iota_i32(h, c.data(), nv, 0);
  iota_i32(h, accepted.data(), nv, 0);
  HIP_TRY(hipMemcpyAsync(a.data(), k.data(), nv * sizeof(double), hipMemcpyDeviceToDevice, h.stream));
  int const vb = bits_of_u((uint64_t)std::max<int64_t>(nv - 1, 1));
  auto modularity = [&]() {  // detail::compute_modularity (common_methods.cuh:176-228)
    hipLaunchKernelGGL(k_sum_internal, 1, 1024, 0, h.stream, (int32_t const*)L.src.data(), (int32_t const*)L.dst.data(), (double const*)L.w.data(),
                       (int32_t const*)c.data(), ne, scal.data());
    hipLaunchKernelGGL(k_sum_squares, 1, 1024, 0, h.stream, (double const*)a.data(), nv, scal.data() + 1);
    double s[2];
    h.read_back(s, scal.data(), 2);
    return s[0] / m - (resolution * s[1]) / (m * m);
  };
  double new_q = modularity();
  double cur_q = new_q - 1.0;
  bool up_down = true;
  double const min_gain = std::max(threshold / (double)std::max<int64_t>(nv, 1), 1e-15);  // compute_louvain_min_vertex_move_gain
  while (new_q > cur_q + threshold) {
    cur_q = new_q;
    // update_clustering_by_delta_modularity (common_methods.cuh:259-447)
    if (ne > 0) {
      hipLaunchKernelGGL(k_pair_keys, g_e, kBlock, 0, h.stream, (int32_t const*)L.src.data(), (int32_t const*)nullptr, (int32_t const*)L.dst.data(),
                         (int32_t const*)c.data(), ne, ekeys.data(), eperm.data());
      sort_pairs(h, ekeys, eperm, ne, vb, vb);
    }
    hipLaunchKernelGGL(k_best_move, g_v, kBlock, 0, h.stream, (int32_t const*)L.off.data(), (uint64_t const*)ekeys.data(), (uint32_t const*)eperm.data(),
                       (int32_t const*)L.dst.data(), (double const*)L.w.data(), (int32_t const*)c.data(), (double const*)k.data(), (double const*)a.data(), m,
                       resolution, nv, best_c.data(), best_d.data());
    HIP_TRY(hipMemsetAsync(count.data(), 0, sizeof(uint32_t), h.stream));
    hipLaunchKernelGGL(k_count_moves, g_v, kBlock, 0, h.stream, (int32_t const*)c.data(), (int32_t const*)best_c.data(), (double const*)best_d.data(), min_gain,
                       up_down ? 1 : 0, nv, count.data());
    uint32_t nr_moves = 0;
    h.read_back(&nr_moves, count.data(), 1);
    if (nr_moves == 0) up_down = !up_down;
    hipLaunchKernelGGL(k_apply_moves, g_v, kBlock, 0, h.stream, c.data(), (int32_t const*)best_c.data(), (double const*)best_d.data(), min_gain, up_down ? 1 : 0, nv);
    // compute_cluster_keys_and_values: cluster weights of the new clustering
    hipLaunchKernelGGL(k_single_keys, g_v, kBlock, 0, h.stream, (int32_t const*)c.data(), nv, vkeys.data(), vperm.data());
    sort_pairs(h, vkeys, vperm, nv, 0, vb);
    hipLaunchKernelGGL(k_heads, g_v, kBlock, 0, h.stream, (uint64_t const*)vkeys.data(), nv, vhead.data());
    HIP_TRY(hipMemsetAsync(a.data(), 0, nv * sizeof(double), h.stream));
    hipLaunchKernelGGL(k_cluster_weights, g_v, kBlock, 0, h.stream, (uint64_t const*)vkeys.data(), (uint32_t const*)vperm.data(), (uint32_t const*)vhead.data(),
                       (double const*)k.data(), nv, a.data());
    up_down = !up_down;
    new_q   = modularity();
    if (new_q > cur_q + threshold) HIP_TRY(hipMemcpyAsync(accepted.data(), c.data(), nv * sizeof(int32_t), hipMemcpyDeviceToDevice, h.stream));
  }
  h.sync();
  return cur_q;
}

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
    graph_t& g        = G(graph);
    CGA_EXPECTS(result != nullptr, CUGRAPH_INVALID_INPUT, "result is NULL");
    HIP_TRY(hipSetDevice(h.device));
    ensure_orientation(h, g, false);  // louvain expects store_transposed == false (louvain.cpp:60-66)
    orientation_t const& o = g.csr;
    int64_t const nv0 = g.nv;
    level_t L;
    L.nv = nv0;
    L.ne = g.ne;
    size_t const e1 = (size_t)std::max<int64_t>(L.ne, 1);
    L.src.resize_discard(e1); L.dst.resize_discard(e1); L.w.resize_discard(e1);
    if (L.ne > 0) {
      hipLaunchKernelGGL(k_expand_src, grid_for(nv0, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)o.offsets.data(), nv0, L.src.data());
      HIP_TRY(hipMemcpyAsync(L.dst.data(), o.indices.data(), L.ne * sizeof(int32_t), hipMemcpyDeviceToDevice, h.stream));
      int const ge = grid_for(L.ne, kBlock, 8192);
      if (!g.has_weights) hipLaunchKernelGGL(k_to_double<float>, ge, kBlock, 0, h.stream, (float const*)nullptr, L.ne, L.w.data());  // constant weight 1 (louvain.cpp:86-92)
      else if (g.weight_type == FLOAT64) hipLaunchKernelGGL(k_to_double<double>, ge, kBlock, 0, h.stream, o.weights.as<double const>(), L.ne, L.w.data());
      else hipLaunchKernelGGL(k_to_double<float>, ge, kBlock, 0, h.stream, o.weights.as<float const>(), L.ne, L.w.data());
    }
    dvec<double> scal(1);
    hipLaunchKernelGGL(k_sum_all, 1, 1024, 0, h.stream, (double const*)L.w.data(), L.ne, scal.data());
    double const m = read_double(h, scal.data());  // compute_total_edge_weight
    auto part      = std::make_unique<device_array_t>((size_t)nv0, g.vertex_type);
    iota_i32(h, part->buf.as<int32_t>(), nv0, 0);
    double best = -1.0;
    size_t levels = 0;
    while (levels < max_level && L.nv > 0 && m > 0.0) {
      ++levels;
      build_offsets(h, L);
      dvec<int32_t> c;
      double const q = run_level(h, L, m, threshold, resolution, c);
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
      hipLaunchKernelGGL(k_compose, grid_for(nv0, kBlock, 8192), kBlock, 0, h.stream, part->buf.as<int32_t>(), (int32_t const*)c.data(), nv0);
      level_t N;
      N.nv = ncl;
      if (L.ne > 0) {
        dvec<uint64_t> keys((size_t)L.ne);
        dvec<uint32_t> perm((size_t)L.ne), head((size_t)L.ne + 1), pos((size_t)L.ne + 1);
        hipLaunchKernelGGL(k_pair_keys, grid_for(L.ne, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)L.src.data(), (int32_t const*)c.data(),
                           (int32_t const*)L.dst.data(), (int32_t const*)c.data(), L.ne, keys.data(), perm.data());
        int const cb = bits_of_u((uint64_t)std::max<int64_t>((int64_t)ncl - 1, 1));
        sort_pairs(h, keys, perm, L.ne, cb, cb);
        hipLaunchKernelGGL(k_heads, grid_for(L.ne, kBlock, 8192), kBlock, 0, h.stream, (uint64_t const*)keys.data(), L.ne, head.data());
        HIP_TRY(hipMemsetAsync(head.data() + L.ne, 0, sizeof(uint32_t), h.stream));
        exclusive_scan_u32(h, head.data(), pos.data(), L.ne + 1);
        uint32_t nce = 0;
        h.read_back(&nce, pos.data() + L.ne, 1);
        N.ne = nce;
        size_t const n1 = (size_t)std::max<uint32_t>(nce, 1);
        N.src.resize_discard(n1); N.dst.resize_discard(n1); N.w.resize_discard(n1);
        hipLaunchKernelGGL(k_coarse_edges, grid_for(L.ne, kBlock, 8192), kBlock, 0, h.stream, (uint64_t const*)keys.data(), (uint32_t const*)perm.data(),
                           (uint32_t const*)head.data(), (uint32_t const*)pos.data(), (double const*)L.w.data(), L.ne, N.src.data(), N.dst.data(), N.w.data());
        h.sync();
      } else {
        N.ne = 0;
        N.src.resize_discard(1); N.dst.resize_discard(1); N.w.resize_discard(1);
      }
      L = std::move(N);
    }
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
