// INT64 and sparse external vertex ids.
//
// The reference instantiates its whole stack for (vertex_t, edge_t) = (int32, int32) and (int64, int64) and renumbers external ids
// through a hash map (cpp/src/c_api/graph_sg.cpp:745-779, cpp/src/utilities/graph_traits.hpp:36-59,
// cpp/src/structure/renumber_utils_impl.cuh:333-660); python-cugraph hands over int64 columns by default.  Here the kernels
// keep 32-bit INTERNAL ids (a graph has fewer than 2^31 vertices and edges per GPU -- the tiled SpMV even addresses edges
// with 16-bit tile-local ids) and the C API translates at its boundary:
//   * graph creation collects the distinct external ids (src, dst, vertices), sorts them (LSD radix sort on the order-preserving
//     64-bit pattern) and replaces every id by its rank -- a "compact" int32 id that is monotone in the external id, so
//     every minimum-external-id tie-break of the algorithms is unchanged; the usual renumbering then runs on compact ids;
//   * ids that come in later (BFS sources, personalization vertices, ...) are looked up by binary search in the sorted list,
//     ids that go out (result vertex columns, predecessors, paths, decompressed edge lists) are gathered from it, and BFS
//     distances are widened to the vertex type.
// The same path serves INT32 graphs whose id range is too sparse for the dense external->internal table (range > 4 x ids).
#include "common.hpp"

namespace cga {
namespace {

__device__ __forceinline__ uint64_t ord_key(int64_t v) { return (uint64_t)v ^ 0x8000000000000000ull; }

template <typename T>
__global__ void k_ids_to_keys(T const* ids, int64_t n, uint64_t* keys)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) keys[i] = ord_key((int64_t)ids[i]);
}
__global__ void k_unique_flags(uint64_t const* keys, int64_t n, uint32_t* flag)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
__global__ void k_unique_emit(uint64_t const* keys, uint32_t const* flag, uint32_t const* pos, int64_t n, int64_t* out)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride)
    if (flag[i]) out[pos[i]] = (int64_t)(keys[i] ^ 0x8000000000000000ull);
}
// rank of ids[i] in the sorted unique list, -1 when absent
template <typename T>
__global__ void k_lookup_sorted(T const* ids, int64_t n, int64_t const* ext, int64_t m, int32_t* out)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    int64_t const x = (int64_t)ids[i];
    int64_t lo = 0, hi = m;
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      if (ext[mid] < x) lo = mid + 1; else hi = mid;
    }
    out[i] = (lo < m && ext[lo] == x) ? (int32_t)lo : -1;
  }
}
template <typename T>
__global__ void k_narrow_ids(T const* ids, int64_t n, int32_t* out)
{  // identity mapping: the id itself when it fits a non-negative int32, else -1 (not a vertex)
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    int64_t const x = (int64_t)ids[i];
    out[i] = (x >= 0 && x < (int64_t)INT32_MAX) ? (int32_t)x : -1;
  }
}
template <typename T>
__global__ void k_from_compact(int32_t const* ids, int64_t n, int64_t const* ext, T* out)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    int32_t const c = ids[i];
    out[i] = c < 0 ? (T)c : (ext ? (T)ext[c] : (T)c);  // negative markers (-1 = none) pass through
  }
}
__global__ void k_widen_dist(int32_t const* d, int64_t n, int64_t* out)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = d[i] == INT32_MAX ? INT64_MAX : (int64_t)d[i];
}
__global__ void k_narrow_dist(int64_t const* d, int64_t n, int32_t* out)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = d[i] >= (int64_t)INT32_MAX ? INT32_MAX : (int32_t)d[i];
}

}  // namespace

// distinct ids of the given columns (INT32 or INT64 views, NULL entries skipped), ascending
void outer_collect(handle_t const& h, device_array_view_t const* const* cols, int ncols, dvec<int64_t>& ext)
{
  int64_t total = 0;
  for (int c = 0; c < ncols; ++c)
    if (cols[c]) total += (int64_t)cols[c]->size;
  ext = dvec<int64_t>();
  if (total == 0) return;
  CGA_EXPECTS(total < ((int64_t)1 << 32), CUGRAPH_ALLOC_ERROR, "too many vertex ids for the id compaction (positions are 32-bit)");
  dvec<uint64_t> keys(total), keys_tmp(total);
  int64_t at = 0;
  for (int c = 0; c < ncols; ++c) {
    if (!cols[c] || cols[c]->size == 0) continue;
    int64_t const n = (int64_t)cols[c]->size;
    int const g     = grid_for(n, kBlock, 8192);
    if (cols[c]->type == INT64) hipLaunchKernelGGL(k_ids_to_keys<int64_t>, g, kBlock, 0, h.stream, cols[c]->as<int64_t const>(), n, keys.data() + at);
    else hipLaunchKernelGGL(k_ids_to_keys<int32_t>, g, kBlock, 0, h.stream, cols[c]->as<int32_t const>(), n, keys.data() + at);
    at += n;
  }
  radix_sort_u64_u32(h, keys.data(), nullptr, keys_tmp.data(), nullptr, total, 0, 64);  // keys only
  keys_tmp = dvec<uint64_t>();
  dvec<uint32_t> flag(total + 1), pos(total + 1);
  hipLaunchKernelGGL(k_unique_flags, grid_for(total, kBlock, 8192), kBlock, 0, h.stream, (uint64_t const*)keys.data(), total, flag.data());
  HIP_TRY(hipMemsetAsync(flag.data() + total, 0, sizeof(uint32_t), h.stream));
  exclusive_scan_u32(h, flag.data(), pos.data(), total + 1);
  uint32_t m = 0;
  h.read_back(&m, pos.data() + total, 1);
  CGA_EXPECTS((int64_t)m < (int64_t)INT32_MAX, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "more than 2^31 - 1 distinct vertex ids: beyond the 32-bit internal ids of this build");
  ext.resize_discard(m);
  hipLaunchKernelGGL(k_unique_emit, grid_for(total, kBlock, 8192), kBlock, 0, h.stream, (uint64_t const*)keys.data(), (uint32_t const*)flag.data(),
                     (uint32_t const*)pos.data(), total, ext.data());
  h.sync();
}

// external ids (view of the graph's OUTER vertex type) -> compact int32 ids (-1 = not a vertex)
void outer_to_compact(handle_t const& h, outer_ids_t const& o, void const* ids, cugraph_data_type_id_t type, int64_t n, int32_t* out)
{
  if (n <= 0) return;
  int const g = grid_for(n, kBlock, 8192);
  if (o.identity) {
    if (type == INT64) hipLaunchKernelGGL(k_narrow_ids<int64_t>, g, kBlock, 0, h.stream, static_cast<int64_t const*>(ids), n, out);
    else hipLaunchKernelGGL(k_narrow_ids<int32_t>, g, kBlock, 0, h.stream, static_cast<int32_t const*>(ids), n, out);
    return;
  }
  int64_t const m = (int64_t)o.ext.size();
  if (type == INT64) hipLaunchKernelGGL(k_lookup_sorted<int64_t>, g, kBlock, 0, h.stream, static_cast<int64_t const*>(ids), n, (int64_t const*)o.ext.data(), m, out);
  else hipLaunchKernelGGL(k_lookup_sorted<int32_t>, g, kBlock, 0, h.stream, static_cast<int32_t const*>(ids), n, (int64_t const*)o.ext.data(), m, out);
}

// compact int32 ids -> new array of the outer vertex type (negative markers pass through)
device_array_t* outer_from_compact(handle_t const& h, outer_ids_t const& o, int32_t const* ids, int64_t n)
{
  auto out = std::make_unique<device_array_t>((size_t)n, o.type);
  if (n > 0) {
    int const g            = grid_for(n, kBlock, 8192);
    int64_t const* const e = o.identity ? nullptr : (int64_t const*)o.ext.data();
    if (o.type == INT64) hipLaunchKernelGGL(k_from_compact<int64_t>, g, kBlock, 0, h.stream, ids, n, e, out->buf.as<int64_t>());
    else hipLaunchKernelGGL(k_from_compact<int32_t>, g, kBlock, 0, h.stream, ids, n, e, out->buf.as<int32_t>());
  }
  return out.release();
}

// in place on an owning array: a column of compact ids becomes a column of outer ids (no-op when the graph has no outer ids)
void outer_replace_ids(handle_t const& h, graph_t const& g, device_array_t*& col)
{
  if (!g.outer.active || col == nullptr) return;
  device_array_t* n = outer_from_compact(h, g.outer, col->buf.as<int32_t>(), (int64_t)col->size);
  h.sync();
  delete col;
  col = n;
}

// BFS distances are typed like the vertices (bfs.cpp:156-187): INT32 hop counts -> INT64 (unreached = type max)
void outer_replace_dist(handle_t const& h, graph_t const& g, device_array_t*& col)
{
  if (!g.outer.active || g.outer.type != INT64 || col == nullptr) return;
  auto out = std::make_unique<device_array_t>(col->size, INT64);
  if (col->size > 0) hipLaunchKernelGGL(k_widen_dist, grid_for((int64_t)col->size, kBlock, 8192), kBlock, 0, h.stream, col->buf.as<int32_t const>(), (int64_t)col->size, out->buf.as<int64_t>());
  h.sync();
  delete col;
  col = out.release();
}

void outer_narrow_dist(handle_t const& h, int64_t const* d, int64_t n, int32_t* out)
{
  if (n > 0) hipLaunchKernelGGL(k_narrow_dist, grid_for(n, kBlock, 8192), kBlock, 0, h.stream, d, n, out);
}

}  // namespace cga
