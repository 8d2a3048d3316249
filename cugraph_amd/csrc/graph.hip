// Graph construction for the PageRank / BFS / SSSP path, written for gfx950.
//
// Replaces:
//   cugraph_graph_create_sg / _with_times_sg / _from_csr     cpp/src/c_api/graph_sg.cpp:699, :835, :989
//   create_graph_from_edgelist (SG)                          cpp/src/structure/create_graph_from_edgelist_impl.cuh:1434-1686
//   renumber_edgelist / compute_renumber_map                 cpp/src/structure/renumber_edgelist_impl.cuh:425-829
//   sort_and_compress_edgelist                               cpp/src/structure/detail/structure_utils.cuh:197-464
//   transpose_graph_storage (as "build the other orientation under the SAME numbering")
//                                                            cpp/src/structure/transpose_graph_storage_impl.cuh:41-100
//   renumber_ext_vertices / unrenumber_int_vertices          cpp/src/structure/renumber_utils_impl.cuh:333-660
//   cugraph_has_vertex                                       cpp/src/c_api/graph_functions.cpp:391
//
// Design (MI355X-first, not the reference's): external ids are mapped through a dense HBM-resident table
// (no hash map: 288 GB of HBM makes a 4 B x id-range table the cheapest ext->int map for analytics inputs);
// vertices are renumbered by descending major degree with ties by ascending external id (deterministic,
// where the reference's thrust::sort_by_key is unstable, renumber_edgelist_impl.cuh:734-738); COO -> CSR/CSC
// is one stable LSD radix sort of packed (major << 32 | minor) keys with the edge position as payload;
// CSR and CSC coexist under one numbering.
#include "common.hpp"
#include "mg_graph.hpp"

namespace cga {

namespace {

__global__ void k_mark(int32_t const* ids, int64_t n, int64_t vmin, uint32_t* flags)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) flags[(int64_t)ids[i] - vmin] = 1u;
}


__global__ void k_degree_keys(uint32_t const* deg, int64_t n, uint32_t maxdeg, uint64_t* keys, uint32_t* vals)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) { keys[i] = (uint64_t)(maxdeg - deg[i]); vals[i] = (uint32_t)i; }
}

// ext_of_compact[rank[r]] = r for every present r
__global__ void k_compact_ext(uint32_t const* flags, uint32_t const* rank, int64_t range, int32_t* ext_of_compact, int64_t vmin)
{
  int64_t r      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; r < range; r += stride)
    if (flags[r]) ext_of_compact[rank[r]] = (int32_t)(r + vmin);
}

// number_map[i] = ext_of_compact[order[i]];  int_of_compact[order[i]] = i
__global__ void k_number_map(uint32_t const* order, int32_t const* ext_of_compact, int64_t nv, int32_t* number_map, int32_t* int_of_compact)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nv; i += stride) {
    uint32_t c        = order[i];
    number_map[i]     = ext_of_compact[c];
    int_of_compact[c] = (int32_t)i;
  }
}

__global__ void k_ext2int(uint32_t const* flags, uint32_t const* rank, int32_t const* int_of_compact, int64_t range, int32_t* ext2int)
{
  int64_t r      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; r < range; r += stride) ext2int[r] = flags[r] ? int_of_compact[rank[r]] : -1;
}

// ids[i] <- table[ids[i] - vmin] (or -1 when out of the table); table == nullptr: identity with bound nv
__global__ void k_lookup(int32_t* ids, int64_t n, int32_t const* table, int64_t vmin, int64_t range, int64_t nv)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    int64_t v = ids[i];
    if (table) {
      int64_t r = v - vmin;
      ids[i]    = (r >= 0 && r < range) ? table[r] : -1;
    } else {
      ids[i] = (v >= 0 && v < nv) ? (int32_t)v : -1;
    }
  }
}

__global__ void k_unrenumber(int32_t* ids, int64_t n, int32_t const* number_map)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    int32_t v = ids[i];
    if (v >= 0) ids[i] = number_map[v];
  }
}

// key = major << vb | minor (vb = bits of V - 1): the 2 * vb significant bits are contiguous, so ONE radix sort over them
// orders the edges by (major, minor) -- 7 passes at RMAT-26 instead of 2 x 4 on the two 32-bit halves.  vals (the edge's
// input position) is only carried when weights have to follow the edges.
__global__ void k_pack_keys(int32_t const* major, int32_t const* minor, int64_t n, int vb, uint64_t* keys, uint32_t* vals)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    keys[i] = ((uint64_t)(uint32_t)major[i] << vb) | (uint32_t)minor[i];
    if (vals) vals[i] = (uint32_t)i;
  }
}

// keys are sorted by (major, minor): indices = minor column; offsets[v] = first position whose major >= v
// (row boundaries are detected between neighbouring keys -- no atomics, hub rows cost nothing extra)
__global__ void k_unpack_minor(uint64_t const* keys, int64_t n, int vb, int32_t* indices, int32_t* offsets)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  uint64_t const minor_mask = (1ull << vb) - 1ull;
  for (; i < n; i += stride) {
    uint64_t k  = keys[i];
    indices[i]  = (int32_t)(uint32_t)(k & minor_mask);
    int64_t maj = (int64_t)(k >> vb);
    int64_t prv = i > 0 ? (int64_t)(keys[i - 1] >> vb) : -1;
    for (int64_t v = prv + 1; v <= maj; ++v) offsets[v] = (int32_t)i;  // rows after the last major keep the pre-filled n
  }
}

__global__ void k_row_degrees(int32_t const* offsets, int64_t nv, uint32_t* deg)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < nv; i += stride) deg[i] = (uint32_t)(offsets[i + 1] - offsets[i]);
}

// sorted[i] = deg[order ? order[i] : i]; checks monotone non-increasing; counts rows >= thresholds
__global__ void k_schedule_stats(uint32_t const* deg, int32_t const* order, int64_t nv, unsigned long long* seg /*[4]*/,
                                 uint32_t* not_sorted, uint32_t* maxdeg)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned c[orientation_t::n_seg] = {0, 0, 0, 0, 0};
  uint32_t mx = 0;
  bool bad    = false;
  for (; i < nv; i += stride) {
    uint32_t d = deg[order ? order[i] : i];
    if (i + 1 < nv) {
      uint32_t dn = deg[order ? order[i + 1] : i + 1];
      bad |= dn > d;
    }
    mx = max(mx, d);
#pragma unroll
    for (int k = 0; k < orientation_t::n_seg; ++k) c[k] += d >= (uint32_t)kSegThreshold[k];
  }
#pragma unroll
  for (int k = 0; k < orientation_t::n_seg; ++k) {
    unsigned v = c[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&seg[k], (unsigned long long)v);
  }
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0) atomicMax(maxdeg, mx);
  if (bad) atomicOr(not_sorted, 1u);
}

// major id of every edge position: rows[e] = v for the edges of row v (either row form: the walk is over the STORED rows)
__global__ void k_expand_rows(rows_view_t rv, int32_t* rows)
{
  // one wave per row chunk: rows are short on average, long rows are striped across the wave
  int64_t wave   = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int lane       = threadIdx.x & 63;
  for (int64_t k = wave; k < rv.n_stored; k += nwaves) {
    uint32_t const b = (uint32_t)rv.offsets[k], len = (uint32_t)rv.offsets[k + 1] - b;  // unsigned positions: up to 2^32 - 1 edges
    int32_t const v  = rv.row_of(k);
    for (uint32_t p = lane; p < len; p += 64) rows[b + p] = v;
  }
}

// ---- hypersparse rows (hypersparse_t, common.hpp; the reference's compress_hypersparse_offsets, structure_utils.cuh:139-195)
// keep[i] = 1 when row first + i has an edge (i < nv - first); keep[nv - first] = 0 closes the scan
__global__ void k_dcs_flags(int32_t const* offsets, int64_t first, int64_t nv, uint32_t* keep)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i <= nv - first; i += stride) keep[i] = (i < nv - first && offsets[first + i + 1] != offsets[first + i]) ? 1u : 0u;
}
// rows below `first` copy their offset; a kept row r writes (r, offsets[r]) at its rank; the thread of i == n - first writes the closing offset
__global__ void k_dcs_compact(int32_t const* offsets, uint32_t const* keep, uint32_t const* rank, int64_t first, int64_t nv, int32_t* nzd, int32_t* out)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i <= nv; i += stride) {
    if (i < first) { out[i] = offsets[i]; continue; }
    int64_t const j = i - first;
    if (i == nv) out[first + rank[j]] = offsets[nv];
    else if (keep[j]) { nzd[rank[j]] = (int32_t)i; out[first + rank[j]] = offsets[i]; }
  }
}
// plain offsets back from the hybrid: stored row k's range starts at offsets[k]; a row that is not stored is empty and starts where the next stored
// row starts.  One thread per stored row fills its own entry and the entries of the unstored rows in front of it.
__global__ void k_dcs_inflate(rows_view_t rv, int64_t nv, int32_t* out)
{
  int64_t k      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; k <= rv.n_stored; k += stride) {
    int64_t const r    = k < rv.n_stored ? (int64_t)rv.row_of(k) : nv;
    int64_t const prev = k > rv.first ? (int64_t)rv.row_of(k - 1) : rv.first - 1;  // (rows below first are all stored)
    int32_t const off  = rv.offsets[k];
    if (k < rv.first) { out[k] = off; continue; }
    for (int64_t v = prev + 1; v <= r; ++v) out[v] = off;
  }
}

__global__ void k_has_vertex(int32_t const* ids, int64_t n, int32_t const* table, int64_t vmin, int64_t range, int64_t nv, uint8_t* out)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    int64_t v = ids[i];
    bool ok;
    if (table) {
      int64_t r = v - vmin;
      ok        = (r >= 0 && r < range) && table[r] >= 0;
    } else {
      ok = v >= 0 && v < nv;
    }
    out[i] = ok ? 1 : 0;
  }
}

int bits_for(uint64_t max_value)
{
  int b = 0;
  while (b < 64 && (max_value >> b) != 0) ++b;
  return b < 1 ? 1 : b;
}

// Builds one orientation from an internal-id COO list.  major/minor/weights are left untouched.
struct edge_props_in {  // optional per-edge columns that travel with the edges (same order as major / minor)
  void const* ids{nullptr};
  size_t ids_size{0};  // 4 or 8
  int32_t const* types{nullptr};
};
void build_orientation(handle_t const& h, int64_t nv, int64_t ne, int32_t const* major, int32_t const* minor,
                       void const* weights, size_t wsize, orientation_t& o, edge_props_in const& props = edge_props_in{})
{
  build_trace tr(h, "orientation");
  o.offsets.resize_discard(nv + 1);
  o.indices.resize_discard(ne + kEdgePad);
  HIP_TRY(hipMemsetAsync(o.indices.data() + ne, 0, kEdgePad * sizeof(int32_t), h.stream));
  fill_i32(h, o.offsets.data(), nv + 1, (int32_t)ne);
  if (ne > 0) {
    dvec<uint64_t> keys(ne), keys_tmp(ne);
    bool const payload = weights != nullptr || props.ids != nullptr || props.types != nullptr;  // the sort carries the edge's input position
    dvec<uint32_t> vals(payload ? ne : 0), vals_tmp(payload ? ne : 0);
    uint32_t* const vp  = payload ? vals.data() : nullptr;
    uint32_t* const vtp = payload ? vals_tmp.data() : nullptr;
    int const vb = bits_for(nv > 0 ? (uint64_t)(nv - 1) : 0);  // <= 31
    hipLaunchKernelGGL(k_pack_keys, grid_for(ne, kBlock, 8192), kBlock, 0, h.stream, major, minor, ne, vb, keys.data(), vp);
    tr.step("pack keys");
    radix_sort_u64_u32(h, keys.data(), vp, keys_tmp.data(), vtp, ne, 0, 2 * vb);
    tr.step("sort");
    hipLaunchKernelGGL(k_unpack_minor, grid_for(ne, kBlock, 8192), kBlock, 0, h.stream, (uint64_t const*)keys.data(), ne, vb,
                       o.indices.data(), o.offsets.data());
    if (weights) {
      o.weights.alloc((ne + kEdgePad) * wsize);
      HIP_TRY(hipMemsetAsync(static_cast<char*>(o.weights.ptr) + ne * wsize, 0, kEdgePad * wsize, h.stream));
      if (wsize == 4) gather_b32(h, (uint32_t const*)weights, vals.data(), o.weights.as<uint32_t>(), ne);
      else            gather_b64(h, (uint64_t const*)weights, vals.data(), o.weights.as<uint64_t>(), ne);
    }
    if (props.ids) {  // edge ids / types follow the same permutation (graph_sg.cpp:803-830 keeps them as edge properties)
      o.edge_ids.alloc((size_t)ne * props.ids_size);
      if (props.ids_size == 4) gather_b32(h, (uint32_t const*)props.ids, vals.data(), o.edge_ids.as<uint32_t>(), ne);
      else                     gather_b64(h, (uint64_t const*)props.ids, vals.data(), o.edge_ids.as<uint64_t>(), ne);
    }
    if (props.types) {
      o.edge_types.resize_discard((size_t)ne);
      gather_b32(h, (uint32_t const*)props.types, vals.data(), reinterpret_cast<uint32_t*>(o.edge_types.data()), ne);
    }
    h.sync();  // temporaries die here
  }
  tr.step("unpack (+ weights)");

  // degree-descending row schedule + class boundaries
  dvec<uint32_t> deg(nv > 0 ? nv : 1);
  dvec<unsigned long long> seg(orientation_t::n_seg);
  dvec<uint32_t> flags(2);  // [not_sorted, maxdeg]
  auto stats = [&](int32_t const* order, unsigned long long* seg_h, uint32_t* flags_h) {
    HIP_TRY(hipMemsetAsync(seg.data(), 0, sizeof(unsigned long long) * orientation_t::n_seg, h.stream));
    HIP_TRY(hipMemsetAsync(flags.data(), 0, 2 * sizeof(uint32_t), h.stream));
    if (nv > 0)
      hipLaunchKernelGGL(k_schedule_stats, grid_for(nv, kBlock, 2048), kBlock, 0, h.stream, (uint32_t const*)deg.data(), order, nv,
                         seg.data(), flags.data(), flags.data() + 1);
    h.read_back(seg_h, seg.data(), orientation_t::n_seg);
    h.read_back(flags_h, flags.data(), 2);
  };
  if (nv > 0) hipLaunchKernelGGL(k_row_degrees, grid_for(nv, kBlock, 4096), kBlock, 0, h.stream, (int32_t const*)o.offsets.data(), nv, deg.data());
  unsigned long long seg_h[orientation_t::n_seg];
  uint32_t flags_h[2];
  stats(nullptr, seg_h, flags_h);
  o.max_degree = (int32_t)flags_h[1];
  o.row_order  = dvec<int32_t>();
  if (flags_h[0] != 0) {  // ids are not degree-sorted: build the permutation
    dvec<uint64_t> keys(nv), keys_tmp(nv);
    dvec<uint32_t> vals(nv), vals_tmp(nv);
    hipLaunchKernelGGL(k_degree_keys, grid_for(nv, kBlock, 4096), kBlock, 0, h.stream, (uint32_t const*)deg.data(), nv, flags_h[1], keys.data(), vals.data());
    radix_sort_u64_u32(h, keys.data(), vals.data(), keys_tmp.data(), vals_tmp.data(), nv, 0, bits_for(flags_h[1]));
    o.row_order.resize_discard(nv);
    HIP_TRY(hipMemcpyAsync(o.row_order.data(), vals.data(), nv * sizeof(int32_t), hipMemcpyDeviceToDevice, h.stream));
    h.sync();
  }
  tr.step("row schedule");
  for (int k = 0; k < orientation_t::n_seg; ++k) o.seg[k] = (int64_t)seg_h[k];  // counts do not depend on the order
  o.built = true;
  h.sync();
}

void check_view(device_array_view_t const* v, char const* name)
{
  CGA_EXPECTS(v == nullptr || v->size == 0 || v->data != nullptr, CUGRAPH_INVALID_INPUT, std::string("Invalid input arguments: ") + name + " has a NULL pointer.");
}

cugraph_error_code_t create_sg(cugraph_resource_handle_t const* handle, cugraph_graph_properties_t const* properties,
                               device_array_view_t const* vertices, device_array_view_t const* src,
                               device_array_view_t const* dst, device_array_view_t const* weights,
                               device_array_view_t const* edge_ids, device_array_view_t const* edge_type_ids,
                               device_array_view_t const* t0, device_array_view_t const* t1, bool_t store_transposed,
                               bool_t renumber, bool_t drop_self_loops, bool_t drop_multi_edges, bool_t symmetrize,
                               bool_t do_expensive_check, cugraph_graph_t** graph, cugraph_error_t** error)
{
  if (graph) *graph = nullptr;
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    CGA_EXPECTS(graph != nullptr && properties != nullptr && src != nullptr && dst != nullptr, CUGRAPH_INVALID_INPUT,
                "Invalid input arguments: NULL graph / properties / src / dst.");
    HIP_TRY(hipSetDevice(h.device));
    if (symmetrize == TRUE)
      CGA_EXPECTS(properties->is_symmetric == TRUE, CUGRAPH_INVALID_INPUT,
                  "Invalid input arguments: The graph property must be symmetric if 'symmetrize' is set to True.");
    CGA_EXPECTS(src->size == dst->size, CUGRAPH_INVALID_INPUT, "Invalid input arguments: src size != dst size.");
    CGA_EXPECTS(weights == nullptr || weights->size == src->size, CUGRAPH_INVALID_INPUT,
                "Invalid input arguments: src size != weights size.");
    CGA_EXPECTS(edge_ids == nullptr || edge_ids->size == src->size, CUGRAPH_INVALID_INPUT,
                "Invalid input arguments: src size != edge id prop size");
    CGA_EXPECTS(edge_type_ids == nullptr || edge_type_ids->size == src->size, CUGRAPH_INVALID_INPUT,
                "Invalid input arguments: src size != edge type prop size");
    // type rules (graph_sg.cpp:745-779): vertex columns are INT32 or INT64; a mix promotes the graph to INT64.  INT64 ids and
    // INT32 ids too sparse for the dense external->internal table are translated at the API boundary (outer_ids.hip); the
    // kernels keep 32-bit internal ids (fewer than 2^31 vertices) and UNSIGNED 32-bit edge positions: up to kMaxGraphEdges edges.
    // (The reference caps an INT32 graph at 2^31 - 1 edges because its edge_t is the vertex type, graph_sg.cpp:766-770, and
    // switches to int64 offsets beyond; here graphs of 2^31 .. 2^32 - 4097 edges -- symmetrised RMAT-26, the Graph500 input of
    // scale 26 -- are built and traversed with the same 32-bit arrays; the algorithms that address edges through signed or
    // 16-bit-tiled positions, PageRank and Louvain, refuse such a graph with CUGRAPH_UNSUPPORTED_TYPE_COMBINATION.)
    auto is_id_type = [](device_array_view_t const* v) { return v == nullptr || v->type == INT32 || v->type == INT64; };
    CGA_EXPECTS(is_id_type(src) && is_id_type(dst) && is_id_type(vertices), CUGRAPH_UNSUPPORTED_TYPE_COMBINATION,
                "vertex ids must be INT32 or INT64");
    bool const any64 = src->type == INT64 || dst->type == INT64 || (vertices && vertices->type == INT64);
    CGA_EXPECTS((int64_t)src->size <= kMaxGraphEdges, any64 ? CUGRAPH_UNSUPPORTED_TYPE_COMBINATION : CUGRAPH_INVALID_INPUT,
                "Number of edges won't fit the 32-bit unsigned edge positions of this build (at most 2^32 - 4097 edges)");
    check_view(src, "src"); check_view(dst, "dst"); check_view(vertices, "vertices");
    outer_ids_t outer;
    dvec<int32_t> c_src, c_dst, c_vtx;  // compact ids when the external ids are translated
    device_array_view_t v_src{nullptr, 0, INT32}, v_dst{nullptr, 0, INT32}, v_vtx{nullptr, 0, INT32};
    {
      bool sparse32 = false;
      if (!any64 && renumber == TRUE) {  // dense table = 4 bytes per id of the RANGE: fine up to a few times the number of ids
        int64_t lo = INT64_MAX, hi = INT64_MIN;
        for (device_array_view_t const* v : {src, dst, vertices}) {
          if (!v || v->size == 0) continue;
          int32_t a, b;
          minmax_i32(h, v->as<int32_t>(), (int64_t)v->size, &a, &b);
          lo = std::min<int64_t>(lo, a); hi = std::max<int64_t>(hi, b);
        }
        int64_t const ids = 2 * (int64_t)src->size + (vertices ? (int64_t)vertices->size : 0);
        sparse32 = hi >= lo && ((hi - lo + 1) > 4 * ids + 65536 || (hi - lo + 1) > ((int64_t)1 << 31) - 2);
      }
      if (any64 || sparse32) {
        outer.active   = true;
        outer.type     = any64 ? INT64 : INT32;
        outer.identity = renumber != TRUE;
        if (!outer.identity) {
          device_array_view_t const* cols[3] = {src, dst, vertices};
          outer_collect(h, cols, 3, outer.ext);
        }
        auto conv = [&](device_array_view_t const* v, dvec<int32_t>& owned, device_array_view_t& view) -> device_array_view_t const* {
          if (!v) return nullptr;
          owned.resize_discard(v->size > 0 ? v->size : 1);
          outer_to_compact(h, outer, v->data, v->type, (int64_t)v->size, owned.data());
          view = device_array_view_t{owned.data(), v->size, INT32};
          return &view;
        };
        src      = conv(src, c_src, v_src);
        dst      = conv(dst, c_dst, v_dst);
        vertices = conv(vertices, c_vtx, v_vtx);
        h.sync();
      }
    }
    CGA_EXPECTS(weights == nullptr || weights->type == FLOAT32 || weights->type == FLOAT64,
                CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "weights must be FLOAT32 or FLOAT64");
    // Edge ids / types / start and end times are edge PROPERTIES that only the sampling and lookup families read
    // (graph_sg.cpp:803-830 stores them next to the weights).  None of the algorithms of this library reads them; edge ids and edge
    // type ids are CARRIED through the build's sort permutation (round 4: they used to be dropped silently) and come back from
    // cugraph_decompress_to_edgelist; start / end times are validated (graph_sg.cpp:781-801) and not kept (no accessor of this
    // library returns them).  Together with a flag that rewrites the edge list (drop_self_loops / drop_multi_edges / symmetrize) they
    // are refused: the rewritten list has no one-to-one relation to the input columns.
    CGA_EXPECTS(edge_type_ids == nullptr || edge_type_ids->type == INT32, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "edge type ids must be INT32");
    CGA_EXPECTS(edge_ids == nullptr || edge_ids->type == INT32 || edge_ids->type == INT64, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION,
                "edge ids must be INT32 or INT64");
    CGA_EXPECTS((t0 == nullptr || t0->size == src->size) && (t1 == nullptr || t1->size == src->size), CUGRAPH_INVALID_INPUT,
                "Invalid input arguments: src size != edge time prop size");
    CGA_EXPECTS((t0 == nullptr || t0->type == INT32 || t0->type == INT64) && (t1 == nullptr || t1->type == INT32 || t1->type == INT64) &&
                  (t0 == nullptr || t1 == nullptr || t0->type == t1->type),
                CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "edge start / end times must share one integer type");
    check_view(weights, "weights");

    auto g              = std::make_unique<graph_t>();
    g->outer            = std::move(outer);
    g->vertex_type      = INT32;
    g->edge_type        = INT32;
    g->weight_type      = weights ? weights->type : FLOAT32;
    g->has_weights      = weights != nullptr;
    g->store_transposed = store_transposed == TRUE;
    g->renumbered       = renumber == TRUE;
    g->props            = *properties;
    int64_t const ne    = (int64_t)src->size;
    g->ne               = ne;
    size_t const wsize  = weights ? dtype_size(weights->type) : 0;

    // inputs are borrowed: work on copies (graph_sg.cpp:98-183)
    edge_list_t el;
    el.n = ne;
    el.wsize = wsize;
    el.s.resize_discard(ne > 0 ? ne : 1); el.d.resize_discard(ne > 0 ? ne : 1);
    if (ne > 0) {
      HIP_TRY(hipMemcpyAsync(el.s.data(), src->data, ne * 4, hipMemcpyDeviceToDevice, h.stream));
      HIP_TRY(hipMemcpyAsync(el.d.data(), dst->data, ne * 4, hipMemcpyDeviceToDevice, h.stream));
    }
    bool const preprocess = drop_self_loops == TRUE || drop_multi_edges == TRUE || symmetrize == TRUE;
    if (preprocess && weights && ne > 0) {  // the flags rewrite the edge list: the weights need an owned copy too
      el.w.alloc(ne * wsize);
      HIP_TRY(hipMemcpyAsync(el.w.ptr, weights->data, ne * wsize, hipMemcpyDeviceToDevice, h.stream));
    }
    int64_t const nvl = vertices ? (int64_t)vertices->size : 0;

    int32_t vmin = 0, vmax = -1;
    {
      int32_t a, b;
      bool any = false;
      auto upd = [&](int32_t const* p, int64_t n) {
        if (n <= 0) return;
        minmax_i32(h, p, n, &a, &b);
        if (!any) { vmin = a; vmax = b; any = true; } else { vmin = std::min(vmin, a); vmax = std::max(vmax, b); }
      };
      upd(el.s.data(), ne); upd(el.d.data(), ne);
      if (vertices) upd(vertices->as<int32_t>(), nvl);
    }

    if (preprocess && ne > 0) {  // graph_sg.cpp:185-248: self-loops, then multi-edges, then symmetrize
      int64_t const vrange = (int64_t)vmax - vmin + 1;
      if (drop_self_loops == TRUE) edgelist_drop_self_loops(h, el);
      if (drop_multi_edges == TRUE) edgelist_drop_multi_edges(h, el, vmin, vrange);
      if (symmetrize == TRUE) edgelist_symmetrize(h, el, vmin, vrange);
      CGA_EXPECTS(el.n <= kMaxGraphEdges, CUGRAPH_INVALID_INPUT, "Number of edges won't fit the 32-bit unsigned edge positions of this build");
    }
    int64_t const ne_in = ne;
    if (do_expensive_check == TRUE) {  // create_graph_from_edgelist_impl.cuh:72-126, 1466-1494
      int64_t const vrange = vmax >= vmin ? (int64_t)vmax - vmin + 1 : 0;
      if (vertices && nvl > 0) {
        CGA_EXPECTS(!vertex_list_has_duplicates(h, vertices->as<int32_t>(), nvl, vmin, vrange), CUGRAPH_INVALID_INPUT,
                    "Invalid input argument: vertices should not have duplicates.");
        if (renumber != TRUE) {
          int32_t a = 0, b = -1;
          minmax_i32(h, vertices->as<int32_t>(), nvl, &a, &b);
          CGA_EXPECTS(a == 0 && (int64_t)b == nvl - 1, CUGRAPH_INVALID_INPUT,
                      "Invalid input argument: vertex IDs should be consecutive integers starting from 0 if renumber is false.");
        }
      }
      if (properties->is_symmetric == TRUE)
        CGA_EXPECTS(edgelist_is_symmetric(h, el, vmin, vrange), CUGRAPH_INVALID_INPUT,
                    "Invalid input arguments: graph_properties.is_symmetric is true but the input edge list is not symmetric.");
      if (properties->is_multigraph != TRUE)
        CGA_EXPECTS(!edgelist_has_parallel_edges(h, el, vmin, vrange), CUGRAPH_INVALID_INPUT,
                    "Invalid input arguments: graph_properties.is_multigraph is false but the input edge list has parallel edges.");
    }
    dvec<int32_t>& s = el.s;
    dvec<int32_t>& d = el.d;
    void const* wptr = weights ? (preprocess && ne_in > 0 ? el.w.ptr : weights->data) : nullptr;
    int64_t const ne2 = el.n;
    g->ne = ne2;

    if (renumber == TRUE) {
      int64_t range = vmax >= vmin ? (int64_t)vmax - vmin + 1 : 0;
      CGA_EXPECTS(range <= ((int64_t)1 << 31) - 2, CUGRAPH_NOT_IMPLEMENTED, "external vertex id range too wide for the dense renumbering table");
      build_trace tr(h, "renumber");
      dvec<uint32_t> flags(range + 1), rank(range + 1);
      HIP_TRY(hipMemsetAsync(flags.data(), 0, (range + 1) * 4, h.stream));
      uint32_t nv32 = 0;
      // the vertex list first: when it already covers every id of [vmin, vmax] (the usual dense case) the two random-store
      // passes over the edge endpoints (27 ms each at RMAT-26) cannot add a vertex and are skipped
      if (nvl > 0) {
        hipLaunchKernelGGL(k_mark, grid_for(nvl, kBlock, 8192), kBlock, 0, h.stream, vertices->as<int32_t>(), nvl, (int64_t)vmin, flags.data());
        exclusive_scan_u32(h, flags.data(), rank.data(), range + 1);
        h.read_back(&nv32, rank.data() + range, 1);
      }
      if ((int64_t)nv32 != range) {
        if (ne2 > 0) {
          hipLaunchKernelGGL(k_mark, grid_for(ne2, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)s.data(), ne2, (int64_t)vmin, flags.data());
          hipLaunchKernelGGL(k_mark, grid_for(ne2, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)d.data(), ne2, (int64_t)vmin, flags.data());
        }
        exclusive_scan_u32(h, flags.data(), rank.data(), range + 1);
        h.read_back(&nv32, rank.data() + range, 1);
      }
      int64_t const nv = nv32;
      g->nv            = nv;
      tr.step("vertex set");
      // major degree per compact id
      dvec<uint32_t> deg(nv > 0 ? nv : 1);
      HIP_TRY(hipMemsetAsync(deg.data(), 0, (nv > 0 ? nv : 1) * 4, h.stream));
      int32_t const* major_ext = store_transposed == TRUE ? d.data() : s.data();
      histogram_i32_mapped(h, major_ext, ne2, (int64_t)vmin, (uint32_t const*)rank.data(), deg.data(), range);
      tr.step("major degrees");
      int32_t dmin = 0, dmax = 0;
      if (nv > 0) minmax_i32(h, reinterpret_cast<int32_t const*>(deg.data()), nv, &dmin, &dmax);
      dvec<uint64_t> keys(nv > 0 ? nv : 1), keys_tmp(nv > 0 ? nv : 1);
      dvec<uint32_t> order(nv > 0 ? nv : 1), order_tmp(nv > 0 ? nv : 1);
      dvec<int32_t> ext_of_compact(nv > 0 ? nv : 1), int_of_compact(nv > 0 ? nv : 1);
      g->number_map.resize_discard(nv);
      g->ext2int.resize_discard(range);
      g->ext_min = vmin;
      if (nv > 0) {
        hipLaunchKernelGGL(k_degree_keys, grid_for(nv, kBlock, 4096), kBlock, 0, h.stream, (uint32_t const*)deg.data(), nv, (uint32_t)dmax, keys.data(), order.data());
        radix_sort_u64_u32(h, keys.data(), order.data(), keys_tmp.data(), order_tmp.data(), nv, 0, bits_for((uint32_t)dmax));
        hipLaunchKernelGGL(k_compact_ext, grid_for(range, kBlock, 8192), kBlock, 0, h.stream, (uint32_t const*)flags.data(), (uint32_t const*)rank.data(), range, ext_of_compact.data(), (int64_t)vmin);
        hipLaunchKernelGGL(k_number_map, grid_for(nv, kBlock, 4096), kBlock, 0, h.stream, (uint32_t const*)order.data(), (int32_t const*)ext_of_compact.data(), nv, g->number_map.data(), int_of_compact.data());
        hipLaunchKernelGGL(k_ext2int, grid_for(range, kBlock, 8192), kBlock, 0, h.stream, (uint32_t const*)flags.data(), (uint32_t const*)rank.data(), (int32_t const*)int_of_compact.data(), range, g->ext2int.data());
        tr.step("degree order + maps");
        if (ne2 > 0) {
          hipLaunchKernelGGL(k_lookup, grid_for(ne2, kBlock, 8192), kBlock, 0, h.stream, s.data(), ne2, (int32_t const*)g->ext2int.data(), (int64_t)vmin, range, nv);
          hipLaunchKernelGGL(k_lookup, grid_for(ne2, kBlock, 8192), kBlock, 0, h.stream, d.data(), ne2, (int32_t const*)g->ext2int.data(), (int64_t)vmin, range, nv);
        }
      }
      tr.step("relabel endpoints");
      h.sync();
    } else {
      // ids are 0..V-1, V = |vertices| or max id + 1 (create_graph_from_edgelist_impl.cuh:1519-1521)
      CGA_EXPECTS(vmax < vmin || vmin >= 0, CUGRAPH_INVALID_INPUT, "Invalid input arguments: negative vertex id with renumber = FALSE.");
      int64_t nv = vertices ? nvl : (int64_t)vmax + 1;
      CGA_EXPECTS((int64_t)vmax < nv, CUGRAPH_INVALID_INPUT, "Invalid input arguments: vertex id out of range with renumber = FALSE.");
      g->nv = nv;
      g->number_map.resize_discard(nv);
      iota_i32(h, g->number_map.data(), nv, 0);
    }

    orientation_t& primary = g->store_transposed ? g->csc : g->csr;
    int32_t const* major   = g->store_transposed ? d.data() : s.data();
    int32_t const* minor   = g->store_transposed ? s.data() : d.data();
    edge_props_in props;
    if (edge_ids || edge_type_ids) {
      CGA_EXPECTS(!preprocess, CUGRAPH_NOT_IMPLEMENTED,
                  "edge ids / edge type ids together with drop_self_loops / drop_multi_edges / symmetrize are not supported: the rewritten edge list has no one-to-one relation to them");
      if (edge_ids) { props.ids = edge_ids->data; props.ids_size = dtype_size(edge_ids->type); g->has_edge_ids = true; g->edge_id_type = edge_ids->type; }
      if (edge_type_ids) { props.types = edge_type_ids->as<int32_t const>(); g->has_edge_types = true; }
    }
    build_orientation(h, g->nv, ne2, major, minor, wptr, wsize, primary, props);
    *graph = reinterpret_cast<cugraph_graph_t*>(g.release());
  });
}

}  // namespace

// Builds the missing orientation under the same numbering (the reference re-creates and re-numbers the
// graph instead, cpp/src/c_api/graph.hpp:84-143).
void compress_hypersparse(handle_t const& h, orientation_t& o, int64_t nv, int64_t first)
{
  CGA_EXPECTS(o.built, CUGRAPH_UNKNOWN_ERROR, "compress_hypersparse: the orientation is not built");
  if (o.dcs.active()) return;
  first = std::max<int64_t>(0, std::min(first, nv));
  int64_t const m = nv - first;  // candidate rows
  dvec<uint32_t> keep((size_t)m + 1), rank((size_t)m + 1);
  hipLaunchKernelGGL(k_dcs_flags, grid_for(m + 1, kBlock, 4096), kBlock, 0, h.stream, (int32_t const*)o.offsets.data(), first, nv, keep.data());
  exclusive_scan_u32(h, keep.data(), rank.data(), m + 1);
  uint32_t n_nzd = 0;
  h.read_back(&n_nzd, rank.data() + m, 1);
  hypersparse_t d;
  d.first = first;
  d.n_nzd = (int64_t)n_nzd;
  d.nzd.resize_discard((size_t)std::max<int64_t>(d.n_nzd, 1));
  d.offsets.resize_discard((size_t)(first + d.n_nzd + 1));
  hipLaunchKernelGGL(k_dcs_compact, grid_for(nv + 1, kBlock, 4096), kBlock, 0, h.stream, (int32_t const*)o.offsets.data(), (uint32_t const*)keep.data(),
                     (uint32_t const*)rank.data(), first, nv, d.nzd.data(), d.offsets.data());
  h.sync();
  o.dcs     = std::move(d);
  o.offsets = dvec<int32_t>();  // the hybrid form replaces the plain one
}

void inflate_offsets(handle_t const& h, orientation_t& o, int64_t nv)
{
  if (!o.dcs.active()) return;
  o.offsets.resize_discard((size_t)nv + 1);
  rows_view_t const rv = rows_view(o, nv);
  hipLaunchKernelGGL(k_dcs_inflate, grid_for(rv.n_stored + 1, kBlock, 4096), kBlock, 0, h.stream, rv, nv, o.offsets.data());
  h.sync();
  o.dcs = hypersparse_t{};
}

void ensure_orientation(handle_t const& h, graph_t& g, bool transposed, bool dcs_aware)
{
  orientation_t& want = transposed ? g.csc : g.csr;
  if (want.built) {
    if (!dcs_aware) inflate_offsets(h, want, g.nv);
    return;
  }
  orientation_t& have = transposed ? g.csr : g.csc;
  CGA_EXPECTS(have.built, CUGRAPH_UNKNOWN_ERROR, "graph has no storage");
  dvec<int32_t> rows(g.ne > 0 ? g.ne : 1);
  if (g.nv > 0 && g.ne > 0)
    hipLaunchKernelGGL(k_expand_rows, grid_for(g.nv * 16, kBlock, 8192), kBlock, 0, h.stream, rows_view(have, g.nv), rows.data());
  size_t wsize = g.has_weights ? dtype_size(g.weight_type) : 0;
  // new major = old minor (indices), new minor = old major (rows)
  edge_props_in props;
  if (g.has_edge_ids) { props.ids = have.edge_ids.ptr; props.ids_size = dtype_size(g.edge_id_type); }
  if (g.has_edge_types) props.types = have.edge_types.data();
  build_orientation(h, g.nv, g.ne, have.indices.data(), rows.data(), g.has_weights ? have.weights.ptr : nullptr, wsize, want, props);
}

void renumber_ext_to_int(handle_t const& h, graph_t const& g, int32_t* ids, int64_t n)
{
  if (n <= 0) return;
  hipLaunchKernelGGL(k_lookup, grid_for(n, kBlock, 4096), kBlock, 0, h.stream, ids, n,
                     g.renumbered ? (int32_t const*)g.ext2int.data() : (int32_t const*)nullptr, g.ext_min, (int64_t)g.ext2int.size(), g.nv);
}

void unrenumber_int_to_ext(handle_t const& h, graph_t const& g, int32_t* ids, int64_t n)
{
  if (n <= 0 || !g.renumbered) return;
  hipLaunchKernelGGL(k_unrenumber, grid_for(n, kBlock, 4096), kBlock, 0, h.stream, ids, n, (int32_t const*)g.number_map.data());
}

}  // namespace cga

using namespace cga;

extern "C" cugraph_error_code_t cugraph_graph_create_sg(
  const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
  const cugraph_type_erased_device_array_view_t* vertices, const cugraph_type_erased_device_array_view_t* src,
  const cugraph_type_erased_device_array_view_t* dst, const cugraph_type_erased_device_array_view_t* weights,
  const cugraph_type_erased_device_array_view_t* edge_ids, const cugraph_type_erased_device_array_view_t* edge_type_ids,
  bool_t store_transposed, bool_t renumber, bool_t drop_self_loops, bool_t drop_multi_edges, bool_t symmetrize,
  bool_t do_expensive_check, cugraph_graph_t** graph, cugraph_error_t** error)
{
  return create_sg(handle, properties, V(vertices), V(src), V(dst), V(weights), V(edge_ids), V(edge_type_ids), nullptr, nullptr,
                   store_transposed, renumber, drop_self_loops, drop_multi_edges, symmetrize, do_expensive_check, graph, error);
}

extern "C" cugraph_error_code_t cugraph_graph_create_with_times_sg(
  const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
  const cugraph_type_erased_device_array_view_t* vertices, const cugraph_type_erased_device_array_view_t* src,
  const cugraph_type_erased_device_array_view_t* dst, const cugraph_type_erased_device_array_view_t* weights,
  const cugraph_type_erased_device_array_view_t* edge_ids, const cugraph_type_erased_device_array_view_t* edge_type_ids,
  const cugraph_type_erased_device_array_view_t* edge_start_time_ids,
  const cugraph_type_erased_device_array_view_t* edge_end_time_ids, bool_t store_transposed, bool_t renumber,
  bool_t drop_self_loops, bool_t drop_multi_edges, bool_t symmetrize, bool_t do_expensive_check,
  cugraph_graph_t** graph, cugraph_error_t** error)
{
  return create_sg(handle, properties, V(vertices), V(src), V(dst), V(weights), V(edge_ids), V(edge_type_ids),
                   V(edge_start_time_ids), V(edge_end_time_ids), store_transposed, renumber, drop_self_loops,
                   drop_multi_edges, symmetrize, do_expensive_check, graph, error);
}

// cugraph_graph_create_mg / _with_times_mg (cpp/src/c_api/graph_mg.cpp:326-560; pylibcugraph MGGraph, graphs.pyx:357-700): every
// rank passes ITS slice of the edge list as `num_arrays` arrays of views and the reference shuffles edges to their owners
// (graph_mg.cpp:140).  On a handle created on a communicator (cugraph_amd_comm_create, the stand-in for the reference's raft::handle_t with
// NCCL comms) the call is COLLECTIVE and yields one graph partitioned over the ranks (mg_graph.hip); on a plain one-rank handle --
// what the reference's single-process C tests and pylibcugraph's MGGraph with a one-rank communicator use -- the arrays are concatenated
// and the graph is built as cugraph_graph_create_sg does, always renumbered as an MG graph is (graph_mg.cpp:214).
namespace cga {
namespace {
struct concat_t {
  dev_buf buf;
  device_array_view_t view{nullptr, 0, INT32};
  bool present{false};
};
void concat_views(handle_t const& h, cugraph_type_erased_device_array_view_t const* const* arr, size_t num_arrays, char const* what, concat_t& out)
{
  if (arr == nullptr) return;
  size_t total = 0;
  cugraph_data_type_id_t type = INT32;
  bool any = false;
  for (size_t i = 0; i < num_arrays; ++i) {
    auto v = V(arr[i]);
    if (!v) continue;
    CGA_EXPECTS(!any || v->type == type, CUGRAPH_INVALID_INPUT, std::string("Invalid input arguments: all ") + what + " arrays must have the same type.");
    type = v->type; any = true;
    total += v->size;
  }
  if (!any) return;
  size_t const esz = dtype_size(type);
  out.buf.alloc(std::max<size_t>(total, 1) * esz);
  size_t off = 0;
  for (size_t i = 0; i < num_arrays; ++i) {
    auto v = V(arr[i]);
    if (!v || v->size == 0) continue;
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(out.buf.ptr) + off * esz, v->data, v->size * esz, hipMemcpyDeviceToDevice, h.stream));
    off += v->size;
  }
  h.sync();
  out.view    = device_array_view_t{out.buf.ptr, total, type};
  out.present = true;
}
cugraph_error_code_t create_mg(const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
                               cugraph_type_erased_device_array_view_t const* const* vertices, cugraph_type_erased_device_array_view_t const* const* src,
                               cugraph_type_erased_device_array_view_t const* const* dst, cugraph_type_erased_device_array_view_t const* const* weights,
                               cugraph_type_erased_device_array_view_t const* const* edge_ids, cugraph_type_erased_device_array_view_t const* const* edge_type_ids,
                               cugraph_type_erased_device_array_view_t const* const* t0, cugraph_type_erased_device_array_view_t const* const* t1,
                               bool_t store_transposed, size_t num_arrays, bool_t drop_self_loops, bool_t drop_multi_edges, bool_t symmetrize,
                               bool_t do_expensive_check, cugraph_graph_t** graph, cugraph_error_t** error)
{
  if (graph) *graph = nullptr;
  concat_t cv, cs, cd, cw, ci, ct, c0, c1;
  bool multi = false;
  cugraph_error_code_t rc = guarded(error, [&] {
    handle_t const& h = H(handle);
    multi             = h.comm != nullptr;  // a handle created on a communicator (extensions.h): the collective path, also with one rank
    CGA_EXPECTS(multi || h.comm_size == 1, CUGRAPH_INVALID_HANDLE, "cugraph_graph_create_mg: a multi-rank handle without a communicator");
    CGA_EXPECTS(src != nullptr && dst != nullptr && num_arrays >= 1, CUGRAPH_INVALID_INPUT, "Invalid input arguments: src / dst arrays.");
    HIP_TRY(hipSetDevice(h.device));
    concat_views(h, vertices, num_arrays, "vertices", cv);
    concat_views(h, src, num_arrays, "src", cs);
    concat_views(h, dst, num_arrays, "dst", cd);
    concat_views(h, weights, num_arrays, "weights", cw);
    concat_views(h, edge_ids, num_arrays, "edge_ids", ci);
    concat_views(h, edge_type_ids, num_arrays, "edge_type_ids", ct);
    concat_views(h, t0, num_arrays, "edge_start_time_ids", c0);
    concat_views(h, t1, num_arrays, "edge_end_time_ids", c1);
    CGA_EXPECTS(cs.present && cd.present, CUGRAPH_INVALID_INPUT, "Invalid input arguments: src / dst arrays.");
    if (multi) {
      // collective: this rank's slice becomes part of ONE graph partitioned over the communicator (graph_mg.cpp:140 shuffles the edges
      // to their owners at this point; here the owners depend on the algorithm family and are chosen on first use: mg_graph.hpp)
      CGA_EXPECTS(graph != nullptr && properties != nullptr, CUGRAPH_INVALID_INPUT, "Invalid input arguments: NULL graph / properties.");
      CGA_EXPECTS(cs.view.size == cd.view.size && (!cw.present || cw.view.size == cs.view.size), CUGRAPH_INVALID_INPUT, "Invalid input arguments: src size != dst / weights size.");
      if (symmetrize == TRUE)  // graph_mg.cpp:117-121
        CGA_EXPECTS(properties->is_symmetric == TRUE, CUGRAPH_INVALID_INPUT,
                    "Invalid input arguments: The graph property must be symmetric if 'symmetrize' is set to True.");
      // INT64 vertex ids (graph_mg.cpp:127-151 instantiates vertex_t = int64_t; python-cugraph hands over int64 columns by default): the ranks
      // agree on ONE ascending list of the distinct ids (mg_graph.hip: mg_outer_ids, collective), the columns become compact int32 ids
      // (position in that list: monotone in the external id, so every minimum-external-id tie-break is unchanged) and every vertex column
      // that leaves the library goes back through outer_ids.hip -- the same boundary translation as on one GPU.
      outer_ids_t outer;
      dvec<int32_t> s32, d32, v32;
      device_array_view_t sview = cs.view, dview = cd.view, vview = cv.view;
      if (mg_outer_ids(h, cv.present ? &cv.view : nullptr, &cs.view, &cd.view, outer)) {
        auto narrow = [&](device_array_view_t& v, dvec<int32_t>& owned) {
          owned.resize_discard(std::max<size_t>(v.size, 1));
          outer_to_compact(h, outer, v.data, v.type, (int64_t)v.size, owned.data());
          v = device_array_view_t{owned.data(), v.size, INT32};
        };
        narrow(sview, s32);
        narrow(dview, d32);
        if (cv.present) narrow(vview, v32);
        h.sync();
      }
      auto g              = std::make_unique<graph_t>();
      g->outer            = std::move(outer);
      g->vertex_type      = INT32;
      g->edge_type        = INT32;
      g->weight_type      = cw.present ? cw.view.type : FLOAT32;
      g->has_weights      = cw.present;
      g->store_transposed = store_transposed == TRUE;
      g->renumbered       = true;  // graph_mg.cpp:214
      g->props            = *properties;
      mg_graph_create(h, *g, cv.present ? &vview : nullptr, &sview, &dview, cw.present ? &cw.view : nullptr, ci.present ? &ci.view : nullptr,
                      ct.present ? &ct.view : nullptr, drop_self_loops == TRUE, drop_multi_edges == TRUE, symmetrize == TRUE);
      *graph = reinterpret_cast<cugraph_graph_t*>(g.release());
    }
  });
  if (rc != CUGRAPH_SUCCESS || multi) return rc;
  auto opt = [](concat_t& c) -> device_array_view_t const* { return c.present ? &c.view : nullptr; };
  return create_sg(handle, properties, opt(cv), &cs.view, &cd.view, opt(cw), opt(ci), opt(ct), opt(c0), opt(c1), store_transposed, TRUE,
                   drop_self_loops, drop_multi_edges, symmetrize, do_expensive_check, graph, error);
}
}  // namespace
}  // namespace cga

extern "C" cugraph_error_code_t cugraph_graph_create_mg(
  cugraph_resource_handle_t const* handle, cugraph_graph_properties_t const* properties,
  cugraph_type_erased_device_array_view_t const* const* vertices, cugraph_type_erased_device_array_view_t const* const* src,
  cugraph_type_erased_device_array_view_t const* const* dst, cugraph_type_erased_device_array_view_t const* const* weights,
  cugraph_type_erased_device_array_view_t const* const* edge_ids, cugraph_type_erased_device_array_view_t const* const* edge_type_ids,
  bool_t store_transposed, size_t num_arrays, bool_t drop_self_loops, bool_t drop_multi_edges, bool_t symmetrize, bool_t do_expensive_check,
  cugraph_graph_t** graph, cugraph_error_t** error)
{
  return create_mg(handle, properties, vertices, src, dst, weights, edge_ids, edge_type_ids, nullptr, nullptr, store_transposed, num_arrays,
                   drop_self_loops, drop_multi_edges, symmetrize, do_expensive_check, graph, error);
}

extern "C" cugraph_error_code_t cugraph_graph_create_with_times_mg(
  cugraph_resource_handle_t const* handle, cugraph_graph_properties_t const* properties,
  cugraph_type_erased_device_array_view_t const* const* vertices, cugraph_type_erased_device_array_view_t const* const* src,
  cugraph_type_erased_device_array_view_t const* const* dst, cugraph_type_erased_device_array_view_t const* const* weights,
  cugraph_type_erased_device_array_view_t const* const* edge_ids, cugraph_type_erased_device_array_view_t const* const* edge_type_ids,
  cugraph_type_erased_device_array_view_t const* const* edge_start_time_ids,
  cugraph_type_erased_device_array_view_t const* const* edge_end_time_ids, bool_t store_transposed, size_t num_arrays, bool_t drop_self_loops,
  bool_t drop_multi_edges, bool_t symmetrize, bool_t do_expensive_check, cugraph_graph_t** graph, cugraph_error_t** error)
{
  return create_mg(handle, properties, vertices, src, dst, weights, edge_ids, edge_type_ids, edge_start_time_ids, edge_end_time_ids,
                   store_transposed, num_arrays, drop_self_loops, drop_multi_edges, symmetrize, do_expensive_check, graph, error);
}

extern "C" cugraph_error_code_t cugraph_graph_create_sg_from_csr(
  const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
  const cugraph_type_erased_device_array_view_t* offsets, const cugraph_type_erased_device_array_view_t* indices,
  const cugraph_type_erased_device_array_view_t* weights, const cugraph_type_erased_device_array_view_t* edge_ids,
  const cugraph_type_erased_device_array_view_t* edge_type_ids, bool_t store_transposed, bool_t renumber,
  bool_t symmetrize, bool_t do_expensive_check, cugraph_graph_t** graph, cugraph_error_t** error)
{
  // CSR rows are sources (graph_sg.cpp:989-1095 decompresses to an edge list the same way).
  if (graph) *graph = nullptr;
  dvec<int32_t> rows;
  device_array_view_t rows_view{nullptr, 0, INT32};
  cugraph_error_code_t rc = guarded(error, [&] {
    handle_t const& h = H(handle);
    auto off          = V(offsets);
    auto idx          = V(indices);
    CGA_EXPECTS(off != nullptr && idx != nullptr && off->size >= 1, CUGRAPH_INVALID_INPUT, "Invalid input arguments: offsets / indices.");
    CGA_EXPECTS(off->type == INT32 && idx->type == INT32, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION,
                "this build supports INT32 offsets / indices only");
    int64_t nv = (int64_t)off->size - 1;
    rows.resize_discard(idx->size > 0 ? idx->size : 1);
    if (nv > 0 && idx->size > 0) {
      rows_view_t rv;  // the caller's CSR: plain rows
      rv.offsets = off->as<int32_t>(); rv.first = nv; rv.n_stored = nv;
      hipLaunchKernelGGL(k_expand_rows, grid_for(nv * 16, kBlock, 8192), kBlock, 0, h.stream, rv, rows.data());
    }
    h.sync();
    rows_view = device_array_view_t{rows.data(), idx->size, INT32};
  });
  if (rc != CUGRAPH_SUCCESS) return rc;
  // the vertex list is 0..nv-1 so isolated trailing vertices survive
  dvec<int32_t> verts;
  device_array_view_t verts_view{nullptr, 0, INT32};
  rc = guarded(error, [&] {
    handle_t const& h = H(handle);
    int64_t nv        = (int64_t)V(offsets)->size - 1;
    verts.resize_discard(nv > 0 ? nv : 1);
    iota_i32(h, verts.data(), nv, 0);
    h.sync();
    verts_view = device_array_view_t{verts.data(), (size_t)nv, INT32};
  });
  if (rc != CUGRAPH_SUCCESS) return rc;
  return create_sg(handle, properties, &verts_view, &rows_view, V(indices), V(weights), V(edge_ids), V(edge_type_ids), nullptr,
                   nullptr, store_transposed, renumber, FALSE, FALSE, symmetrize, do_expensive_check, graph, error);
}

extern "C" void cugraph_graph_free(cugraph_graph_t* graph) { delete reinterpret_cast<graph_t*>(graph); }

extern "C" cugraph_error_code_t cugraph_has_vertex(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
                                                   cugraph_type_erased_device_array_view_t* vertices,
                                                   bool_t /*do_expensive_check*/,
                                                   cugraph_type_erased_device_array_t** result, cugraph_error_t** error)
{
  if (result) *result = nullptr;
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    graph_t& g        = GM(graph);
    auto v            = V(vertices);
    CGA_EXPECTS(v != nullptr && result != nullptr, CUGRAPH_INVALID_INPUT, "vertices / result is NULL");
    if (g.mg) {  // multi-GPU graph: the id is a vertex if ANY rank knows it (graph_functions.cpp:391 answers per rank for the local range)
      vertex_column_in c_mv;  // INT64 ids of a multi-GPU graph: compact int32 ids, -1 (never a vertex) when unknown
      v = c_mv.get(h, g, v, "vertices");
      CGA_EXPECTS(v->type == INT32, CUGRAPH_INVALID_INPUT, "vertex type of graph and vertices must match");
      auto out = std::make_unique<device_array_t>(v->size, BOOL);
      mg_has_vertex(h, g, v->as<int32_t>(), (int64_t)v->size, out->buf.as<uint8_t>());
      h.sync();
      *result = reinterpret_cast<cugraph_type_erased_device_array_t*>(out.release());
      return;
    }
    vertex_column_in c_v;  // INT64 / sparse external ids: compact int32 ids (absent ones -1) from here on (outer_ids.hip)
    v = c_v.get(h, g, v, "vertices");
    auto out = std::make_unique<device_array_t>(v->size, BOOL);
    if (v->size > 0)
      hipLaunchKernelGGL(k_has_vertex, grid_for(v->size, kBlock, 4096), kBlock, 0, h.stream, v->as<int32_t>(), (int64_t)v->size,
                         g.renumbered ? (int32_t const*)g.ext2int.data() : (int32_t const*)nullptr, g.ext_min,
                         (int64_t)g.ext2int.size(), g.nv, out->buf.as<uint8_t>());
    h.sync();
    *result = reinterpret_cast<cugraph_type_erased_device_array_t*>(out.release());
  });
}

extern "C" size_t cugraph_amd_graph_num_vertices(const cugraph_graph_t* graph)
{
  return graph ? (size_t) reinterpret_cast<graph_t const*>(graph)->nv : 0;
}
extern "C" size_t cugraph_amd_graph_num_edges(const cugraph_graph_t* graph)
{
  return graph ? (size_t) reinterpret_cast<graph_t const*>(graph)->ne : 0;
}
extern "C" cugraph_error_code_t cugraph_amd_graph_compress_hypersparse(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph, bool_t transposed,
                                                                       size_t first_row, cugraph_error_t** error)
{
  return guarded(error, [&] {
    CGA_EXPECTS(handle != nullptr && graph != nullptr, CUGRAPH_INVALID_INPUT, "handle / graph is NULL");
    handle_t const& h = H(handle);
    graph_t& g        = GM(graph);
    CGA_EXPECTS(!g.mg, CUGRAPH_NOT_IMPLEMENTED, "hypersparse rows: a multi-GPU graph chooses the form of its local partitions itself");
    HIP_TRY(hipSetDevice(h.device));
    ensure_orientation(h, g, transposed == TRUE, /*dcs_aware=*/true);
    orientation_t& o = transposed == TRUE ? g.csc : g.csr;
    CGA_EXPECTS((int64_t)first_row <= g.nv, CUGRAPH_INVALID_INPUT, "hypersparse rows: first_row is larger than the number of vertices");
    if (o.dcs.active() && o.dcs.first != (int64_t)first_row) inflate_offsets(h, o, g.nv);  // another boundary: through the plain form
    compress_hypersparse(h, o, g.nv, (int64_t)first_row);
  });
}

extern "C" cugraph_error_code_t cugraph_amd_graph_hypersparse_view(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph, bool_t transposed,
                                                                   bool_t* is_hypersparse, size_t* first_row, size_t* num_nzd,
                                                                   cugraph_type_erased_device_array_view_t** nzd_rows,
                                                                   cugraph_type_erased_device_array_view_t** offsets, cugraph_error_t** error)
{
  if (nzd_rows) *nzd_rows = nullptr;
  if (offsets) *offsets = nullptr;
  return guarded(error, [&] {
    CGA_EXPECTS(handle != nullptr && graph != nullptr, CUGRAPH_INVALID_INPUT, "handle / graph is NULL");
    handle_t const& h = H(handle);
    graph_t& g        = GM(graph);
    CGA_EXPECTS(!g.mg, CUGRAPH_NOT_IMPLEMENTED, "hypersparse rows: not offered on a multi-GPU graph");
    HIP_TRY(hipSetDevice(h.device));
    ensure_orientation(h, g, transposed == TRUE, /*dcs_aware=*/true);
    orientation_t const& o = transposed == TRUE ? g.csc : g.csr;
    rows_view_t const rv   = rows_view(o, g.nv);
    if (is_hypersparse) *is_hypersparse = rv.plain() ? FALSE : TRUE;
    if (first_row) *first_row = rv.plain() ? 0 : (size_t)rv.first;
    if (num_nzd) *num_nzd = rv.plain() ? 0 : (size_t)(rv.n_stored - rv.first);
    if (nzd_rows)
      *nzd_rows = reinterpret_cast<cugraph_type_erased_device_array_view_t*>(
        new device_array_view_t{const_cast<int32_t*>(rv.nzd), rv.plain() ? (size_t)0 : (size_t)(rv.n_stored - rv.first), INT32});
    if (offsets)
      *offsets = reinterpret_cast<cugraph_type_erased_device_array_view_t*>(new device_array_view_t{const_cast<int32_t*>(rv.offsets), (size_t)rv.n_stored + 1, INT32});
  });
}

// multi-GPU graph: the edges of THIS rank's PageRank partition (0 before the first PageRank call builds it); otherwise all edges
extern "C" size_t cugraph_amd_graph_num_local_edges(const cugraph_graph_t* graph)
{
  if (!graph) return 0;
  graph_t const& g = *reinterpret_cast<graph_t const*>(graph);
  if (!g.mg) return (size_t)g.ne;
  if (g.mg->pr2d) return (size_t)g.mg->pr2d->ne_local;  // (the layout the last PageRank of this graph ran on)
  return g.mg->pr ? (size_t)g.mg->pr->ne_local : 0;
}
