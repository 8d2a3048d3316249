// Edge-list preprocessing behind the graph-creation flags drop_self_loops / drop_multi_edges / symmetrize.
//
// Replaces (call site cpp/src/c_api/graph_sg.cpp:185-248):
//   remove_self_loops        cpp/src/structure/remove_self_loops_impl.cuh
//   remove_multi_edges       cpp/src/structure/remove_multi_edges_impl.cuh  (graph_functions.hpp:1073-1140)
//   symmetrize_edgelist      cpp/src/structure/symmetrize_edgelist_impl.cuh:60-135, 372-1010
// Semantics kept:
//   * multi-edges: one edge per (src, dst) survives -- the minimum-weight one (the reference keeps the minimum when the
//     graph is symmetric and an arbitrary one otherwise; the minimum is a valid instance of "arbitrary");
//   * symmetrize (reciprocal = false): self-loops are kept as they are; for every unordered pair {a > b} the edges a->b
//     ("lower") and b->a ("upper") are each sorted by weight and paired rank by rank: a matched pair becomes ONE undirected
//     edge with the AVERAGE weight, unmatched edges are kept with their own weight, and every resulting edge is emitted
//     in both directions (so the multiplicity of {a,b} is max(#lower, #upper)).
// Everything is sort-based on the hand-written LSD radix sort of prims.hip (no Thrust / hash maps).
#include "common.hpp"

namespace cga {

namespace {

template <typename T>
__global__ void k_gather_t(T const* in, uint32_t const* idx, int64_t n, T* out)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[idx[i]];
}

// out[pos[i]] = in[i] for flagged i
template <typename T>
__global__ void k_compact_t(T const* in, uint32_t const* flag, uint32_t const* pos, int64_t n, T* out)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride)
    if (flag[i]) out[pos[i]] = in[i];
}

__global__ void k_flag_not_self_loop(int32_t const* s, int32_t const* d, int64_t n, uint32_t* flag)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) flag[i] = s[i] != d[i] ? 1u : 0u;
}
__global__ void k_invert_flag(uint32_t* flag, int64_t n)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) flag[i] ^= 1u;
}

__global__ void k_iota_u32(uint32_t* p, int64_t n)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = (uint32_t)i;
}

// order-preserving key of a floating-point weight (ascending)
__device__ __forceinline__ uint64_t weight_key(float w)
{
  uint32_t b = __float_as_uint(w);
  return (uint64_t)((b & 0x80000000u) ? ~b : (b | 0x80000000u));
}
__device__ __forceinline__ uint64_t weight_key(double w)
{
  unsigned long long b = (unsigned long long)__double_as_longlong(w);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
template <typename WT>
__global__ void k_weight_keys(WT const* w, uint32_t const* idx, int64_t n, uint64_t* keys)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) keys[i] = weight_key(w[idx[i]]);
}

// (src, dst) key of edge idx[i]; UNDIRECTED: (max, min) and up[] = 1 when src < dst
template <bool UNDIRECTED>
__global__ void k_pair_keys(int32_t const* s, int32_t const* d, uint32_t const* idx, int64_t n, int64_t vmin, uint64_t* keys)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t e = idx[i];
    uint64_t a = (uint64_t)((int64_t)s[e] - vmin), b = (uint64_t)((int64_t)d[e] - vmin);
    if (UNDIRECTED && a < b) { uint64_t t = a; a = b; b = t; }
    keys[i] = (a << 32) | b;
  }
}
__global__ void k_upper_keys(int32_t const* s, int32_t const* d, uint32_t const* idx, int64_t n, uint64_t* keys)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) { uint32_t e = idx[i]; keys[i] = s[e] < d[e] ? 1ull : 0ull; }  // lower-triangular edges first
}

__global__ void k_flag_heads(uint64_t const* keys, int64_t n, uint32_t* flag)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

__global__ void k_run_index_fix(uint32_t const* head, int64_t n, uint32_t* rid)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) rid[i] = rid[i] + head[i] - 1u;
}

// per run (group of equal undirected pair): first position, first position of its upper-triangular edges
__global__ void k_run_bounds(uint32_t const* head, uint32_t const* rid, uint64_t const* upper /* sorted 0/1 per position */, int64_t n,
                             uint32_t* run_start, uint32_t* upper_start)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t r = rid[i];
    if (head[i]) run_start[r] = (uint32_t)i;
    if (upper[i] && (head[i] || !upper[i - 1])) upper_start[r] = (uint32_t)i;
  }
}

// symmetrize_op_t (symmetrize_edgelist_impl.cuh:78-110), reciprocal = false
template <typename WT>
__global__ void k_symmetrize_select(uint32_t const* rid, uint64_t const* upper, uint32_t const* run_start, uint32_t const* upper_start, int64_t n,
                                    int64_t n_runs, WT const* w_sorted, uint32_t* include, WT* w_out)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t r        = rid[i];
    int64_t const rs  = run_start[r];
    int64_t const re  = (int64_t)r + 1 < n_runs ? (int64_t)run_start[r + 1] : n;
    int64_t const us  = upper_start[r] == 0xFFFFFFFFu ? re : (int64_t)upper_start[r];
    int64_t const nlo = us - rs, nup = re - us;
    if (!upper[i]) {
      int64_t k = i - rs;
      include[i] = 1u;
      if (w_sorted) w_out[i] = k < nup ? (w_sorted[i] + w_sorted[us + k]) / WT(2) : w_sorted[i];
    } else {
      int64_t k = i - us;
      include[i] = k < nlo ? 0u : 1u;  // matched upper edges are represented by their lower partner
      if (w_sorted) w_out[i] = w_sorted[i];
    }
  }
}

// included undirected edge at sorted position i (key = (hi << 32 | lo)) -> two directed edges
template <typename WT>
__global__ void k_symmetrize_emit(uint64_t const* keys, uint32_t const* include, uint32_t const* pos, WT const* w, int64_t n, int64_t m, int64_t vmin,
                                  int32_t* s_out, int32_t* d_out, WT* w_out)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride)
    if (include[i]) {
      int32_t hi = (int32_t)((int64_t)(keys[i] >> 32) + vmin), lo = (int32_t)((int64_t)(keys[i] & 0xFFFFFFFFull) + vmin);
      int64_t p  = pos[i];
      s_out[p] = hi; d_out[p] = lo;
      s_out[m + p] = lo; d_out[m + p] = hi;
      if (w) { w_out[p] = w[i]; w_out[m + p] = w[i]; }
    }
}

int bits_of(uint64_t max_value)
{
  int b = 0;
  while (b < 64 && (max_value >> b) != 0) ++b;
  return b < 1 ? 1 : b;
}

template <typename T>
void compact(handle_t const& h, T const* in, uint32_t const* flag, uint32_t const* pos, int64_t n, T* out)
{
  if (n > 0) hipLaunchKernelGGL(k_compact_t<T>, grid_for(n, kBlock, 8192), kBlock, 0, h.stream, in, flag, pos, n, out);
}

// flags -> (positions, count)
int64_t scan_flags(handle_t const& h, dvec<uint32_t>& flag, dvec<uint32_t>& pos, int64_t n)
{
  HIP_TRY(hipMemsetAsync(flag.data() + n, 0, sizeof(uint32_t), h.stream));
  exclusive_scan_u32(h, flag.data(), pos.data(), n + 1);
  uint32_t total = 0;
  h.read_back(&total, pos.data() + n, 1);
  return (int64_t)total;
}

// stable sort of the edge positions idx[] by the pair key (optionally undirected), weight ascending inside equal keys
template <typename WT>
void sort_edges(handle_t const& h, edge_list_t const& el, int64_t vmin, int64_t vrange, bool undirected, dvec<uint32_t>& idx, dvec<uint64_t>& keys)
{
  int64_t const n = el.n;
  dvec<uint64_t> keys_tmp(n);
  dvec<uint32_t> idx_tmp(n);
  int const g = grid_for(n, kBlock, 8192);
  hipLaunchKernelGGL(k_iota_u32, g, kBlock, 0, h.stream, idx.data(), n);
  if (el.w.ptr) {  // least significant criterion first (LSD)
    hipLaunchKernelGGL(k_weight_keys<WT>, g, kBlock, 0, h.stream, el.w.as<WT const>(), (uint32_t const*)idx.data(), n, keys.data());
    radix_sort_u64_u32(h, keys.data(), idx.data(), keys_tmp.data(), idx_tmp.data(), n, 0, (int)(8 * sizeof(WT)));
  }
  if (undirected) {
    hipLaunchKernelGGL(k_upper_keys, g, kBlock, 0, h.stream, (int32_t const*)el.s.data(), (int32_t const*)el.d.data(), (uint32_t const*)idx.data(), n, keys.data());
    radix_sort_u64_u32(h, keys.data(), idx.data(), keys_tmp.data(), idx_tmp.data(), n, 0, 1);
  }
  int const vb = bits_of(vrange > 0 ? (uint64_t)(vrange - 1) : 0);
  auto pair_keys = [&] {
    if (undirected) hipLaunchKernelGGL(k_pair_keys<true>, g, kBlock, 0, h.stream, (int32_t const*)el.s.data(), (int32_t const*)el.d.data(), (uint32_t const*)idx.data(), n, vmin, keys.data());
    else hipLaunchKernelGGL(k_pair_keys<false>, g, kBlock, 0, h.stream, (int32_t const*)el.s.data(), (int32_t const*)el.d.data(), (uint32_t const*)idx.data(), n, vmin, keys.data());
  };
  pair_keys();
  radix_sort_u64_u32(h, keys.data(), idx.data(), keys_tmp.data(), idx_tmp.data(), n, 0, vb);
  radix_sort_u64_u32(h, keys.data(), idx.data(), keys_tmp.data(), idx_tmp.data(), n, 32, 32 + vb);
  h.sync();
}

template <typename WT>
void drop_multi_edges_t(handle_t const& h, edge_list_t& el, int64_t vmin, int64_t vrange)
{
  int64_t const n = el.n;
  if (n <= 1) return;
  dvec<uint32_t> idx(n), flag(n + 1), pos(n + 1);
  dvec<uint64_t> keys(n);
  sort_edges<WT>(h, el, vmin, vrange, false, idx, keys);
  hipLaunchKernelGGL(k_flag_heads, grid_for(n, kBlock, 8192), kBlock, 0, h.stream, (uint64_t const*)keys.data(), n, flag.data());
  int64_t const m = scan_flags(h, flag, pos, n);
  dvec<int32_t> s2(n), d2(n), s3(m > 0 ? m : 1), d3(m > 0 ? m : 1);
  int const g = grid_for(n, kBlock, 8192);
  hipLaunchKernelGGL(k_gather_t<int32_t>, g, kBlock, 0, h.stream, (int32_t const*)el.s.data(), (uint32_t const*)idx.data(), n, s2.data());
  hipLaunchKernelGGL(k_gather_t<int32_t>, g, kBlock, 0, h.stream, (int32_t const*)el.d.data(), (uint32_t const*)idx.data(), n, d2.data());
  compact(h, (int32_t const*)s2.data(), flag.data(), pos.data(), n, s3.data());
  compact(h, (int32_t const*)d2.data(), flag.data(), pos.data(), n, d3.data());
  if (el.w.ptr) {
    dev_buf w2(n * sizeof(WT)), w3((m > 0 ? m : 1) * sizeof(WT));
    hipLaunchKernelGGL(k_gather_t<WT>, g, kBlock, 0, h.stream, el.w.as<WT const>(), (uint32_t const*)idx.data(), n, w2.as<WT>());
    compact(h, w2.as<WT const>(), flag.data(), pos.data(), n, w3.as<WT>());  // first of each group = minimum weight
    h.sync();
    el.w = std::move(w3);
  }
  h.sync();
  el.s = std::move(s3);
  el.d = std::move(d3);
  el.n = m;
}

template <typename WT>
void symmetrize_t(handle_t const& h, edge_list_t& el, int64_t vmin, int64_t vrange)
{
  int64_t const n0 = el.n;
  if (n0 == 0) return;
  bool const weighted = el.w.ptr != nullptr;
  // 1. separate the self-loops (kept as they are)
  dvec<uint32_t> flag(n0 + 1), pos(n0 + 1);
  int const g0 = grid_for(n0, kBlock, 8192);
  hipLaunchKernelGGL(k_flag_not_self_loop, g0, kBlock, 0, h.stream, (int32_t const*)el.s.data(), (int32_t const*)el.d.data(), n0, flag.data());
  int64_t const n = scan_flags(h, flag, pos, n0);  // off-diagonal edges
  int64_t const nd = n0 - n;
  edge_list_t off;
  off.n = n;
  off.s.resize_discard(n > 0 ? n : 1); off.d.resize_discard(n > 0 ? n : 1);
  compact(h, (int32_t const*)el.s.data(), flag.data(), pos.data(), n0, off.s.data());
  compact(h, (int32_t const*)el.d.data(), flag.data(), pos.data(), n0, off.d.data());
  if (weighted) { off.w.alloc((n > 0 ? n : 1) * sizeof(WT)); compact(h, el.w.as<WT const>(), flag.data(), pos.data(), n0, off.w.as<WT>()); }
  dvec<int32_t> ds(nd > 0 ? nd : 1), dd(nd > 0 ? nd : 1);
  dev_buf dw;
  hipLaunchKernelGGL(k_invert_flag, g0, kBlock, 0, h.stream, flag.data(), n0);
  int64_t const nd_check = scan_flags(h, flag, pos, n0);
  CGA_EXPECTS(nd_check == nd, CUGRAPH_UNKNOWN_ERROR, "symmetrize: self-loop count mismatch");
  compact(h, (int32_t const*)el.s.data(), flag.data(), pos.data(), n0, ds.data());
  compact(h, (int32_t const*)el.d.data(), flag.data(), pos.data(), n0, dd.data());
  if (weighted) { dw.alloc((nd > 0 ? nd : 1) * sizeof(WT)); compact(h, el.w.as<WT const>(), flag.data(), pos.data(), n0, dw.as<WT>()); }
  h.sync();

  // 2. off-diagonal edges ordered by (unordered pair, lower before upper, weight)
  int64_t m = 0;
  dvec<int32_t> so, dox;
  dev_buf wo;
  if (n > 0) {
    dvec<uint32_t> idx(n), head(n + 1), rid(n + 1), include(n + 1), ipos(n + 1);
    dvec<uint64_t> keys(n), upper(n);
    sort_edges<WT>(h, off, vmin, vrange, true, idx, keys);
    int const g = grid_for(n, kBlock, 8192);
    hipLaunchKernelGGL(k_upper_keys, g, kBlock, 0, h.stream, (int32_t const*)off.s.data(), (int32_t const*)off.d.data(), (uint32_t const*)idx.data(), n, upper.data());
    hipLaunchKernelGGL(k_flag_heads, g, kBlock, 0, h.stream, (uint64_t const*)keys.data(), n, head.data());
    int64_t const n_runs = scan_flags(h, head, rid, n);
    // the exclusive prefix counts the heads BEFORE i: a head gets its own run index, the others index + 1
    dvec<uint32_t> run_start(n_runs + 1), upper_start(n_runs + 1);
    fill_u32(h, upper_start.data(), n_runs + 1, 0xFFFFFFFFu);
    hipLaunchKernelGGL(k_run_index_fix, g, kBlock, 0, h.stream, (uint32_t const*)head.data(), n, rid.data());
    hipLaunchKernelGGL(k_run_bounds, g, kBlock, 0, h.stream, (uint32_t const*)head.data(), (uint32_t const*)rid.data(), (uint64_t const*)upper.data(), n,
                       run_start.data(), upper_start.data());
    dev_buf w_sorted, w_sel;
    if (weighted) {
      w_sorted.alloc(n * sizeof(WT)); w_sel.alloc(n * sizeof(WT));
      hipLaunchKernelGGL(k_gather_t<WT>, g, kBlock, 0, h.stream, off.w.as<WT const>(), (uint32_t const*)idx.data(), n, w_sorted.as<WT>());
    }
    hipLaunchKernelGGL(k_symmetrize_select<WT>, g, kBlock, 0, h.stream, (uint32_t const*)rid.data(), (uint64_t const*)upper.data(), (uint32_t const*)run_start.data(),
                       (uint32_t const*)upper_start.data(), n, n_runs, weighted ? w_sorted.as<WT const>() : (WT const*)nullptr, include.data(),
                       weighted ? w_sel.as<WT>() : (WT*)nullptr);
    m = scan_flags(h, include, ipos, n);
    so.resize_discard(2 * m + nd + 1); dox.resize_discard(2 * m + nd + 1);
    if (weighted) wo.alloc((2 * m + nd + 1) * sizeof(WT));
    hipLaunchKernelGGL(k_symmetrize_emit<WT>, g, kBlock, 0, h.stream, (uint64_t const*)keys.data(), (uint32_t const*)include.data(), (uint32_t const*)ipos.data(),
                       weighted ? w_sel.as<WT const>() : (WT const*)nullptr, n, m, vmin, so.data(), dox.data(), weighted ? wo.as<WT>() : (WT*)nullptr);
    h.sync();
  } else {
    so.resize_discard(nd + 1); dox.resize_discard(nd + 1);
    if (weighted) wo.alloc((nd + 1) * sizeof(WT));
  }
  // 3. lower + mirrored upper + self-loops
  if (nd > 0) {
    HIP_TRY(hipMemcpyAsync(so.data() + 2 * m, ds.data(), nd * 4, hipMemcpyDeviceToDevice, h.stream));
    HIP_TRY(hipMemcpyAsync(dox.data() + 2 * m, dd.data(), nd * 4, hipMemcpyDeviceToDevice, h.stream));
    if (weighted) HIP_TRY(hipMemcpyAsync(wo.as<WT>() + 2 * m, dw.ptr, nd * sizeof(WT), hipMemcpyDeviceToDevice, h.stream));
  }
  h.sync();
  el.s = std::move(so);
  el.d = std::move(dox);
  if (weighted) el.w = std::move(wo);
  el.n = 2 * m + nd;
}

}  // namespace

namespace {

__global__ void k_directed_keys(int32_t const* s, int32_t const* d, int64_t n, int64_t vmin, uint64_t* fwd, uint64_t* rev)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t const stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint64_t const a = (uint64_t)((int64_t)s[i] - vmin), b = (uint64_t)((int64_t)d[i] - vmin);
    fwd[i] = (a << 32) | b;
    if (rev) rev[i] = (b << 32) | a;
  }
}
__global__ void k_vertex_keys(int32_t const* v, int64_t n, int64_t vmin, uint64_t* keys)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t const stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) keys[i] = (uint64_t)((int64_t)v[i] - vmin);
}
// *flag = 1 if a[i] != b[i] anywhere (b != nullptr) or a[i] == a[i - 1] anywhere (b == nullptr)
__global__ void k_any_mismatch(uint64_t const* a, uint64_t const* b, int64_t n, uint32_t* flag)
{
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t const stride = (int64_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (; i < n; i += stride) bad |= b ? a[i] != b[i] : (i > 0 && a[i] == a[i - 1]);
  if (bad) *flag = 1u;  // same value from every writer
}

void sort_keys(handle_t const& h, dvec<uint64_t>& keys, int64_t n, int lo_bits, int hi_bits)
{
  dvec<uint64_t> keys_tmp(n);
  dvec<uint32_t> idx(n), idx_tmp(n);
  radix_sort_u64_u32(h, keys.data(), idx.data(), keys_tmp.data(), idx_tmp.data(), n, 0, lo_bits);
  if (hi_bits > 0) radix_sort_u64_u32(h, keys.data(), idx.data(), keys_tmp.data(), idx_tmp.data(), n, 32, 32 + hi_bits);
}

bool any_mismatch(handle_t const& h, uint64_t const* a, uint64_t const* b, int64_t n)
{
  dvec<uint32_t> flag(1);
  HIP_TRY(hipMemsetAsync(flag.data(), 0, sizeof(uint32_t), h.stream));
  hipLaunchKernelGGL(k_any_mismatch, grid_for(n, kBlock, 8192), kBlock, 0, h.stream, a, b, n, flag.data());
  uint32_t f = 0;
  h.read_back(&f, flag.data(), 1);
  return f != 0;
}

}  // namespace

// do_expensive_check (create_graph_from_edgelist_impl.cuh:261-333): the edge multiset equals its transpose
bool edgelist_is_symmetric(handle_t const& h, edge_list_t const& el, int64_t vmin, int64_t vrange)
{
  int64_t const n = el.n;
  if (n == 0) return true;
  int const vb = bits_of(vrange > 0 ? (uint64_t)(vrange - 1) : 0);
  dvec<uint64_t> fwd(n), rev(n);
  hipLaunchKernelGGL(k_directed_keys, grid_for(n, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)el.s.data(), (int32_t const*)el.d.data(), n, vmin, fwd.data(), rev.data());
  sort_keys(h, fwd, n, vb, vb);
  sort_keys(h, rev, n, vb, vb);
  return !any_mismatch(h, fwd.data(), rev.data(), n);
}

bool edgelist_has_parallel_edges(handle_t const& h, edge_list_t const& el, int64_t vmin, int64_t vrange)
{
  int64_t const n = el.n;
  if (n <= 1) return false;
  int const vb = bits_of(vrange > 0 ? (uint64_t)(vrange - 1) : 0);
  dvec<uint64_t> fwd(n);
  hipLaunchKernelGGL(k_directed_keys, grid_for(n, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)el.s.data(), (int32_t const*)el.d.data(), n, vmin, fwd.data(), (uint64_t*)nullptr);
  sort_keys(h, fwd, n, vb, vb);
  return any_mismatch(h, fwd.data(), nullptr, n);
}

bool vertex_list_has_duplicates(handle_t const& h, int32_t const* v, int64_t n, int64_t vmin, int64_t vrange)
{
  if (n <= 1) return false;
  dvec<uint64_t> keys(n);
  hipLaunchKernelGGL(k_vertex_keys, grid_for(n, kBlock, 8192), kBlock, 0, h.stream, v, n, vmin, keys.data());
  sort_keys(h, keys, n, bits_of(vrange > 0 ? (uint64_t)(vrange - 1) : 0), 0);
  return any_mismatch(h, keys.data(), nullptr, n);
}

void edgelist_drop_self_loops(handle_t const& h, edge_list_t& el)
{
  int64_t const n = el.n;
  if (n == 0) return;
  dvec<uint32_t> flag(n + 1), pos(n + 1);
  hipLaunchKernelGGL(k_flag_not_self_loop, grid_for(n, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)el.s.data(), (int32_t const*)el.d.data(), n, flag.data());
  int64_t const m = scan_flags(h, flag, pos, n);
  dvec<int32_t> s2(m > 0 ? m : 1), d2(m > 0 ? m : 1);
  compact(h, (int32_t const*)el.s.data(), flag.data(), pos.data(), n, s2.data());
  compact(h, (int32_t const*)el.d.data(), flag.data(), pos.data(), n, d2.data());
  if (el.w.ptr) {
    dev_buf w2((m > 0 ? m : 1) * el.wsize);
    if (el.wsize == 4) compact(h, el.w.as<uint32_t const>(), flag.data(), pos.data(), n, w2.as<uint32_t>());
    else compact(h, el.w.as<uint64_t const>(), flag.data(), pos.data(), n, w2.as<uint64_t>());
    h.sync();
    el.w = std::move(w2);
  }
  h.sync();
  el.s = std::move(s2);
  el.d = std::move(d2);
  el.n = m;
}

void edgelist_drop_multi_edges(handle_t const& h, edge_list_t& el, int64_t vmin, int64_t vrange)
{
  if (el.w.ptr && el.wsize == 8) drop_multi_edges_t<double>(h, el, vmin, vrange);
  else drop_multi_edges_t<float>(h, el, vmin, vrange);
}

void edgelist_symmetrize(handle_t const& h, edge_list_t& el, int64_t vmin, int64_t vrange)
{
  if (el.w.ptr && el.wsize == 8) symmetrize_t<double>(h, el, vmin, vrange);
  else symmetrize_t<float>(h, el, vmin, vrange);
}

}  // namespace cga
