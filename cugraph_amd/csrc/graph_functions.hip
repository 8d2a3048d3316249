// cugraph_degrees / cugraph_in_degrees / cugraph_out_degrees and cugraph_extract_paths (the calls
// user code makes right after building a graph and right after a BFS).
//
// Replaces:
//   cugraph_{in_,out_,}degrees, cugraph_degrees_result_*   cpp/src/c_api/degrees.cu:24-213, degrees_result.cpp:9-57
//     (graph_view_t::compute_in_degrees / compute_out_degrees, cpp/src/structure/graph_view_impl.cuh:649-690)
//   cugraph_extract_paths, cugraph_extract_paths_result_*  cpp/src/c_api/extract_paths.cpp:24-170
//     (cugraph::extract_bfs_paths, cpp/src/traversal/extract_bfs_paths_impl.cuh:130-240)
// Degrees come straight from the offsets of the orientation whose rows are the wanted endpoint; when only the other
// orientation exists they are a histogram of its minor ids (no transposition of the storage is forced).
#include "common.hpp"
#include "mg_graph.hpp"

#include <climits>

#include "cugraph_c/graph_functions.h"
#include "cugraph_c/traversal_algorithms.h"

namespace cga {

struct degrees_result_t {  // c_api/degrees_result.hpp
  bool is_symmetric{false};
  device_array_t* vertex_ids{nullptr};
  device_array_t* in_degrees{nullptr};
  device_array_t* out_degrees{nullptr};
  ~degrees_result_t() { delete vertex_ids; delete in_degrees; delete out_degrees; }
};

struct extract_paths_result_t {  // c_api/extract_paths.cpp:24-30
  size_t max_path_length{0};
  device_array_t* paths{nullptr};
  ~extract_paths_result_t() { delete paths; }
};

namespace {

// deg[row] of every STORED row (either row form; the rows a hypersparse orientation does not store keep the caller's 0)
__global__ void k_offsets_to_degrees(rows_view_t rv, int32_t* deg)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < rv.n_stored; i += stride) deg[rv.row_of(i)] = (int32_t)((uint32_t)rv.offsets[i + 1] - (uint32_t)rv.offsets[i]);
}

__global__ void k_gather_degrees(int32_t const* deg, int32_t const* internal_ids, int64_t n, int32_t* out)
{
  int64_t i      = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = deg[internal_ids[i]];
}

// degree of every internal vertex as the endpoint `rows_are_sources ? source : destination`
void endpoint_degrees(handle_t const& h, graph_t const& g, bool as_source, int32_t* deg)
{
  int64_t const nv = g.nv;
  if (nv == 0) return;
  orientation_t const& same  = as_source ? g.csr : g.csc;   // rows are the wanted endpoint
  orientation_t const& other = as_source ? g.csc : g.csr;
  if (same.built) {
    rows_view_t const rv = rows_view(same, nv);
    if (!rv.plain()) HIP_TRY(hipMemsetAsync(deg, 0, nv * sizeof(int32_t), h.stream));
    hipLaunchKernelGGL(k_offsets_to_degrees, grid_for(rv.n_stored, kBlock, 8192), kBlock, 0, h.stream, rv, deg);
  } else {
    CGA_EXPECTS(other.built, CUGRAPH_UNKNOWN_ERROR, "graph has no adjacency storage");
    HIP_TRY(hipMemsetAsync(deg, 0, nv * sizeof(int32_t), h.stream));
    if (g.ne > 0) histogram_i32(h, other.indices.data(), g.ne, reinterpret_cast<uint32_t*>(deg), g.nv);
  }
}

cugraph_error_code_t degrees_impl(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
                                  const cugraph_type_erased_device_array_view_t* source_vertices, bool want_in, bool want_out,
                                  cugraph_degrees_result_t** result, cugraph_error_t** error)
{
  if (result) *result = nullptr;
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    CGA_EXPECTS(result != nullptr, CUGRAPH_INVALID_INPUT, "result is NULL");
    HIP_TRY(hipSetDevice(h.device));
    auto const sv_user = reinterpret_cast<device_array_view_t const*>(source_vertices);
    if (GM(graph).mg) {  // a graph from cugraph_graph_create_mg on a communicator handle: collective, every rank answers for its share of the vertices
      graph_t& mgg = GM(graph);
      vertex_column_in c_msv;  // INT64 ids: compact int32 ids from here on (unknown ids -1: refused below like any id that is no vertex)
      device_array_view_t const* msv = nullptr;
      mg_agree(mgg, [&] {  // rank-local argument checks: every rank fails or none does (a rank that threw alone would leave its peers in the collective below)
        msv = c_msv.get(h, mgg, sv_user, "source_vertices");
        CGA_EXPECTS(msv == nullptr || msv->type == INT32, CUGRAPH_INVALID_INPUT, "vertex type of graph and source_vertices must match");
      }, "cugraph_degrees");
      bool const msym   = mgg.props.is_symmetric == TRUE;
      bool const mshare = want_in && want_out && msym;  // degrees.cu:84-88
      dvec<int32_t> ids, din, dout;
      int64_t const n = mg_degrees(h, mgg, msv, want_in, want_out && !mshare, ids, din, dout);
      auto res          = std::make_unique<degrees_result_t>();
      res->is_symmetric = msym;
      auto take = [&](dvec<int32_t> const& d, cugraph_data_type_id_t t) {
        auto* a = new device_array_t((size_t)n, t);
        if (n > 0) HIP_TRY(hipMemcpyAsync(a->buf.ptr, d.data(), (size_t)n * 4, hipMemcpyDeviceToDevice, h.stream));
        return a;
      };
      res->vertex_ids = take(ids, INT32);
      if (want_in) res->in_degrees = take(din, mgg.edge_type);
      if (want_out && !mshare) res->out_degrees = take(dout, mgg.edge_type);
      h.sync();
      outer_replace_ids(h, mgg, res->vertex_ids);
      *result = reinterpret_cast<cugraph_degrees_result_t*>(res.release());
      return;
    }
    graph_t& g = G(graph);
    vertex_column_in c_sv;  // INT64 / sparse external ids: compact int32 ids from here on (outer_ids.hip)
    device_array_view_t const* sv = c_sv.get(h, g, sv_user, "source_vertices");
    int64_t const nv = g.nv, n1 = nv > 0 ? nv : 1;
    bool const sym   = g.props.is_symmetric == TRUE;
    bool const share = want_in && want_out && sym;  // degrees.cu:84-88: one array serves both
    dvec<int32_t> din, dout;
    if (want_in) { din.resize_discard(n1); endpoint_degrees(h, g, false, din.data()); }
    if (want_out && !share) { dout.resize_discard(n1); endpoint_degrees(h, g, true, dout.data()); }
    auto res          = std::make_unique<degrees_result_t>();
    res->is_symmetric = sym;
    int64_t n_out     = nv;
    dvec<int32_t> ids;
    if (sv) {
      n_out = (int64_t)sv->size;
      ids.resize_discard(n_out > 0 ? n_out : 1);
      res->vertex_ids = new device_array_t((size_t)n_out, INT32);
      if (n_out > 0) {
        HIP_TRY(hipMemcpyAsync(ids.data(), sv->data, n_out * 4, hipMemcpyDeviceToDevice, h.stream));
        HIP_TRY(hipMemcpyAsync(res->vertex_ids->buf.ptr, sv->data, n_out * 4, hipMemcpyDeviceToDevice, h.stream));
        renumber_ext_to_int(h, g, ids.data(), n_out);
        CGA_EXPECTS(count_negative_i32(h, ids.data(), n_out) == 0, CUGRAPH_INVALID_INPUT, "Invalid input argument: source_vertices contains a vertex that is not in the graph");
      }
    } else {
      res->vertex_ids = new device_array_t((size_t)nv, INT32);
      if (nv > 0) HIP_TRY(hipMemcpyAsync(res->vertex_ids->buf.ptr, g.number_map.data(), nv * 4, hipMemcpyDeviceToDevice, h.stream));
    }
    auto emit = [&](dvec<int32_t> const& d) {
      auto* a = new device_array_t((size_t)n_out, g.edge_type);
      if (n_out > 0) {
        if (sv) hipLaunchKernelGGL(k_gather_degrees, grid_for(n_out, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)d.data(), (int32_t const*)ids.data(), n_out, a->buf.as<int32_t>());
        else HIP_TRY(hipMemcpyAsync(a->buf.ptr, d.data(), n_out * 4, hipMemcpyDeviceToDevice, h.stream));
      }
      return a;
    };
    if (want_in) res->in_degrees = emit(din);
    if (want_out && !share) res->out_degrees = emit(dout);
    h.sync();
    if (g.outer.active) {  // the vertex column goes back in the caller's id type: its own column, or the mapped number_map
      if (sv_user) {
        delete res->vertex_ids;
        res->vertex_ids = new device_array_t((size_t)n_out, sv_user->type);
        if (n_out > 0) HIP_TRY(hipMemcpyAsync(res->vertex_ids->buf.ptr, sv_user->data, n_out * dtype_size(sv_user->type), hipMemcpyDeviceToDevice, h.stream));
        h.sync();
      } else {
        outer_replace_ids(h, g, res->vertex_ids);
      }
    }
    *result = reinterpret_cast<cugraph_degrees_result_t*>(res.release());
  });
}

// ---- extract_paths
__global__ void k_max_path(int32_t const* dest, int64_t n, int32_t const* dist, int32_t const* pred, int32_t* out_max, int32_t* bad)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t const v = dest[i];
  if (v < 0) { *bad = 1; return; }
  int32_t const d = dist[v];
  if (pred[v] >= 0 && d != INT32_MAX) atomicMax(out_max, d);  // compute_max_distance: 0 for vertices without a predecessor
}

// one thread walks one path back from its destination (BFS depths are tens of hops, so the walk is short and the rows of the
// matrix are written by consecutive threads)
__global__ void k_walk_paths(int32_t const* dest, int64_t n, int32_t const* dist, int32_t const* pred_int, int32_t const* number_map, int64_t L,
                             int32_t* paths)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t v = dest[i];
  int32_t d = dist[v];
  if (d == INT32_MAX || (int64_t)d >= L) return;  // unreached destination: the row stays invalid (-1)
  for (int32_t pos = d; pos >= 0 && v >= 0; --pos) {
    paths[i * L + pos] = number_map[v];
    v                  = pred_int[v];
  }
}

// the same two steps on a multi-GPU graph: the tables cover the dense external id range and hold hop count + 1 / predecessor + 2 (mg_gather_paths)
__global__ void k_mg_max_path(int32_t const* dest, int64_t n, int64_t vmin, int64_t vrange, uint32_t const* dist1, uint32_t const* pred2, int32_t* out_max, int32_t* bad)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t const k = (int64_t)dest[i] - vmin;
  if (k < 0 || k >= vrange || dist1[k] == 0u) { *bad = 1; return; }
  uint32_t const d = dist1[k] - 1u;
  if (pred2[k] >= 2u && d != (uint32_t)INT32_MAX) atomicMax(out_max, (int32_t)d);
}
__global__ void k_mg_walk_paths(int32_t const* dest, int64_t n, int64_t vmin, int64_t vrange, uint32_t const* dist1, uint32_t const* pred2, int64_t L, int32_t* paths)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t k = (int64_t)dest[i] - vmin;
  uint32_t const d = dist1[k] - 1u;
  if (d == (uint32_t)INT32_MAX || (int64_t)d >= L) return;  // unreached destination: the row stays invalid (-1)
  for (int64_t pos = d; pos >= 0; --pos) {
    paths[i * L + pos] = (int32_t)(k + vmin);
    int64_t const p = (int64_t)pred2[k] - 2;
    if (p < 0) break;
    k = p - vmin;
    if (k < 0 || k >= vrange) break;
  }
}

}  // namespace
}  // namespace cga

using namespace cga;

extern "C" cugraph_error_code_t cugraph_in_degrees(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
                                                   const cugraph_type_erased_device_array_view_t* source_vertices, bool_t /*do_expensive_check*/,
                                                   cugraph_degrees_result_t** result, cugraph_error_t** error)
{
  return degrees_impl(handle, graph, source_vertices, true, false, result, error);
}
extern "C" cugraph_error_code_t cugraph_out_degrees(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
                                                    const cugraph_type_erased_device_array_view_t* source_vertices, bool_t /*do_expensive_check*/,
                                                    cugraph_degrees_result_t** result, cugraph_error_t** error)
{
  return degrees_impl(handle, graph, source_vertices, false, true, result, error);
}
extern "C" cugraph_error_code_t cugraph_degrees(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
                                                const cugraph_type_erased_device_array_view_t* source_vertices, bool_t /*do_expensive_check*/,
                                                cugraph_degrees_result_t** result, cugraph_error_t** error)
{
  return degrees_impl(handle, graph, source_vertices, true, true, result, error);
}

static degrees_result_t& DR(cugraph_degrees_result_t* r) { return *reinterpret_cast<degrees_result_t*>(r); }
static cugraph_type_erased_device_array_view_t* as_view(device_array_t* a)
{
  return a ? reinterpret_cast<cugraph_type_erased_device_array_view_t*>(a->new_view()) : nullptr;
}
extern "C" cugraph_type_erased_device_array_view_t* cugraph_degrees_result_get_vertices(cugraph_degrees_result_t* r) { return as_view(DR(r).vertex_ids); }
extern "C" cugraph_type_erased_device_array_view_t* cugraph_degrees_result_get_in_degrees(cugraph_degrees_result_t* r) { return as_view(DR(r).in_degrees); }
extern "C" cugraph_type_erased_device_array_view_t* cugraph_degrees_result_get_out_degrees(cugraph_degrees_result_t* r)
{  // degrees_result.cpp:30-42: a symmetric graph's out-degrees ARE its in-degrees
  degrees_result_t& d = DR(r);
  return d.out_degrees ? as_view(d.out_degrees) : d.is_symmetric ? as_view(d.in_degrees) : nullptr;
}
extern "C" void cugraph_degrees_result_free(cugraph_degrees_result_t* r) { delete reinterpret_cast<degrees_result_t*>(r); }

extern "C" cugraph_error_code_t cugraph_extract_paths(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
                                                      const cugraph_type_erased_device_array_view_t* /*sources*/,
                                                      const cugraph_paths_result_t* paths_result,
                                                      const cugraph_type_erased_device_array_view_t* destinations,
                                                      cugraph_extract_paths_result_t** result, cugraph_error_t** error)
{
  if (result) *result = nullptr;
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    graph_t& g        = GM(graph);
    auto pr           = reinterpret_cast<paths_result_t const*>(paths_result);
    auto dv           = reinterpret_cast<device_array_view_t const*>(destinations);
    auto const local_checks = [&] {
      CGA_EXPECTS(result != nullptr && pr != nullptr && dv != nullptr, CUGRAPH_INVALID_INPUT, "NULL argument");
      CGA_EXPECTS(pr->distances != nullptr && pr->distances->type == g.api_vertex_type(), CUGRAPH_INVALID_INPUT,
                  "Invalid input argument: distances must come from cugraph_bfs (vertex-typed hop counts)");
    };
    if (!g.mg) local_checks();
    if (g.mg) {
      // A graph from cugraph_graph_create_mg (extract_paths.cpp:57-119 with multi_gpu = true; extract_bfs_paths_impl.cuh:130-240 looks the
      // predecessors of vertices other ranks own up through a distributed key-value store, one hop per round).  COLLECTIVE: the BFS
      // result holds this rank's share of the vertices; the (hop count, predecessor) columns of all ranks are folded into two tables over
      // the dense id range that every rank holds (8 bytes per id: what the graph's presence table and degree arrays cost already), and
      // each rank then walks its own destinations locally -- any destination, whoever owns it.  The matrix is as wide on every rank
      // (the longest path over all ranks' destinations), as the reference's host_scalar_allreduce makes it.
      mg_graph_t& mg    = *g.mg;
      HIP_TRY(hipSetDevice(h.device));
      vertex_column_in c_mdv;
      // rank-local argument checks: every rank fails or none does (round 6; a rank that threw alone left its peers in mg_gather_paths until the
      // communicator's timeout)
      mg_agree(g, [&] { local_checks(); dv = c_mdv.get(h, g, dv, "destinations"); }, "cugraph_extract_paths");
      int64_t const n   = (int64_t)pr->vertex_ids->size;
      bool const bad_in = pr->predecessors == nullptr || (int64_t)pr->predecessors->size != n || (int64_t)pr->distances->size != n;
      int64_t const nd = (int64_t)dv->size, n1 = std::max<int64_t>(n, 1);
      dvec<int32_t> vert((size_t)n1), pred((size_t)n1), dist32((size_t)n1), scal(2);
      if (!bad_in && n > 0) {
        if (g.outer.active) {  // the BFS result carries outer ids / widened hop counts: back to compact ids and int32 hops
          outer_to_compact(h, g.outer, pr->vertex_ids->buf.ptr, pr->vertex_ids->type, n, vert.data());
          outer_to_compact(h, g.outer, pr->predecessors->buf.ptr, pr->predecessors->type, n, pred.data());  // -1 stays -1
          if (g.outer.type == INT64) outer_narrow_dist(h, pr->distances->buf.as<int64_t const>(), n, dist32.data());
          else HIP_TRY(hipMemcpyAsync(dist32.data(), pr->distances->buf.ptr, (size_t)n * 4, hipMemcpyDeviceToDevice, h.stream));
        } else {
          HIP_TRY(hipMemcpyAsync(vert.data(), pr->vertex_ids->buf.ptr, (size_t)n * 4, hipMemcpyDeviceToDevice, h.stream));
          HIP_TRY(hipMemcpyAsync(pred.data(), pr->predecessors->buf.ptr, (size_t)n * 4, hipMemcpyDeviceToDevice, h.stream));
          HIP_TRY(hipMemcpyAsync(dist32.data(), pr->distances->buf.ptr, (size_t)n * 4, hipMemcpyDeviceToDevice, h.stream));
        }
      }
      dvec<uint32_t> dist1, pred2;
      mg_gather_paths(h, g, vert.data(), dist32.data(), pred.data(), bad_in ? 0 : n, dist1, pred2);
      HIP_TRY(hipMemsetAsync(scal.data(), 0, 2 * sizeof(int32_t), h.stream));
      if (nd > 0)
        hipLaunchKernelGGL(k_mg_max_path, grid_for(nd), kBlock, 0, h.stream, dv->as<int32_t>(), nd, mg.vmin, mg.vrange, (uint32_t const*)dist1.data(), (uint32_t const*)pred2.data(),
                           scal.data(), scal.data() + 1);
      int32_t hs[2] = {0, 0};
      h.read_back(hs, scal.data(), 2);
      // one exchange carries the width and the verdicts, so that every rank throws or none does
      int64_t const agreed = mg_host_max(g, bad_in ? ((int64_t)1 << 41) : hs[1] != 0 ? ((int64_t)1 << 40) : (int64_t)hs[0]);
      CGA_EXPECTS(agreed < ((int64_t)1 << 41), CUGRAPH_INVALID_INPUT, "Invalid input argument: predecessors cannot be null");  // extract_bfs_paths_impl.cuh:140-142
      CGA_EXPECTS(agreed < ((int64_t)1 << 40), CUGRAPH_INVALID_INPUT, "Invalid input argument: destinations contains a vertex that is not in the graph");
      int64_t const L = agreed + 1;
      auto res        = std::make_unique<extract_paths_result_t>();
      res->max_path_length = (size_t)L;
      res->paths      = new device_array_t((size_t)(nd * L), INT32);
      if (nd > 0) {
        HIP_TRY(hipMemsetAsync(res->paths->buf.ptr, 0xFF, (size_t)(nd * L) * 4, h.stream));  // invalid_vertex_id = -1
        hipLaunchKernelGGL(k_mg_walk_paths, grid_for(nd), kBlock, 0, h.stream, dv->as<int32_t>(), nd, mg.vmin, mg.vrange, (uint32_t const*)dist1.data(), (uint32_t const*)pred2.data(), L,
                           res->paths->buf.as<int32_t>());
      }
      h.sync();
      outer_replace_ids(h, g, res->paths);  // -1 padding passes through
      *result = reinterpret_cast<cugraph_extract_paths_result_t*>(res.release());
      return;
    }
    CGA_EXPECTS(pr->predecessors != nullptr && (int64_t)pr->predecessors->size == g.nv, CUGRAPH_INVALID_INPUT,
                "Invalid input argument: predecessors cannot be null");  // extract_bfs_paths_impl.cuh:140-142
    HIP_TRY(hipSetDevice(h.device));
    vertex_column_in c_dv;  // INT64 / sparse external ids: compact int32 ids from here on (outer_ids.hip)
    dv = c_dv.get(h, g, dv, "destinations");
    int64_t const nv = g.nv, nd = (int64_t)dv->size;
    dvec<int32_t> dest(nd > 0 ? nd : 1), pred(nv > 0 ? nv : 1), scal(2), dist32;
    if (nd > 0) HIP_TRY(hipMemcpyAsync(dest.data(), dv->data, nd * 4, hipMemcpyDeviceToDevice, h.stream));
    int32_t const* dist = pr->distances->buf.as<int32_t>();
    if (g.outer.active) {  // the BFS result carries outer ids / widened hop counts: back to compact ids and int32 hops
      if (nv > 0) outer_to_compact(h, g.outer, pr->predecessors->buf.ptr, pr->predecessors->type, nv, pred.data());  // -1 stays -1
      if (g.outer.type == INT64) {
        dist32.resize_discard(nv > 0 ? nv : 1);
        outer_narrow_dist(h, pr->distances->buf.as<int64_t const>(), nv, dist32.data());
        dist = dist32.data();
      }
    } else if (nv > 0) {
      HIP_TRY(hipMemcpyAsync(pred.data(), pr->predecessors->buf.ptr, nv * 4, hipMemcpyDeviceToDevice, h.stream));
    }
    renumber_ext_to_int(h, g, dest.data(), nd);   // extract_paths.cpp:93-109: destinations and predecessors to internal ids
    renumber_ext_to_int(h, g, pred.data(), nv);   // -1 (no predecessor) stays negative
    HIP_TRY(hipMemsetAsync(scal.data(), 0, 2 * sizeof(int32_t), h.stream));
    if (nd > 0) hipLaunchKernelGGL(k_max_path, grid_for(nd), kBlock, 0, h.stream, (int32_t const*)dest.data(), nd, dist, (int32_t const*)pred.data(), scal.data(), scal.data() + 1);
    int32_t hs[2] = {0, 0};
    h.read_back(hs, scal.data(), 2);
    CGA_EXPECTS(hs[1] == 0, CUGRAPH_INVALID_INPUT, "Invalid input argument: destinations contains a vertex that is not in the graph");
    int64_t const L = (int64_t)hs[0] + 1;  // extract_bfs_paths_impl.cuh:165-173
    auto res        = std::make_unique<extract_paths_result_t>();
    res->max_path_length = (size_t)L;
    res->paths      = new device_array_t((size_t)(nd * L), g.vertex_type);
    if (nd > 0) {
      HIP_TRY(hipMemsetAsync(res->paths->buf.ptr, 0xFF, (size_t)(nd * L) * 4, h.stream));  // invalid_vertex_id = -1
      hipLaunchKernelGGL(k_walk_paths, grid_for(nd), kBlock, 0, h.stream, (int32_t const*)dest.data(), nd, dist, (int32_t const*)pred.data(),
                         (int32_t const*)g.number_map.data(), L, res->paths->buf.as<int32_t>());
    }
    h.sync();
    outer_replace_ids(h, g, res->paths);  // -1 padding passes through
    *result = reinterpret_cast<cugraph_extract_paths_result_t*>(res.release());
  });
}

extern "C" size_t cugraph_extract_paths_result_get_max_path_length(cugraph_extract_paths_result_t* result)
{
  return reinterpret_cast<extract_paths_result_t*>(result)->max_path_length;
}
extern "C" cugraph_type_erased_device_array_view_t* cugraph_extract_paths_result_get_paths(cugraph_extract_paths_result_t* result)
{
  return as_view(reinterpret_cast<extract_paths_result_t*>(result)->paths);
}
extern "C" void cugraph_extract_paths_result_free(cugraph_extract_paths_result_t* result) { delete reinterpret_cast<extract_paths_result_t*>(result); }

// ------------------------------------------------------------------------------------------------------------------
// cugraph_decompress_to_edgelist + cugraph_edgelist_* (cpp/include/cugraph_c/graph_functions.h:399-480, impl
// cpp/src/c_api/decompress_to_edgelist.cpp:24-125 -> cugraph::decompress_to_edgelist, cpp/src/structure/
// decompress_to_edgelist_impl.cuh): the graph back as (sources, destinations[, weights]) with EXTERNAL ids, in the order of
// the by-source storage (the reference flips a transposed graph first, decompress_to_edgelist.cpp:53-58; here the CSR
// orientation is built next to the CSC under the same numbering).  Edge ids / edge type ids given at graph creation travel with
// their edges (graph.hip: build_orientation) and come back here; the offsets accessor (a temporal-graph feature) returns NULL.
namespace cga {
struct edgelist_result_t {
  device_array_t* src{nullptr};
  device_array_t* dst{nullptr};
  device_array_t* wgt{nullptr};
  device_array_t* ids{nullptr};
  device_array_t* types{nullptr};
  ~edgelist_result_t() { delete src; delete dst; delete wgt; delete ids; delete types; }
};
namespace {
__global__ void k_rows_of_edges(rows_view_t rv, int32_t* rows)  // (either row form: the walk is over the stored rows)
{
  int64_t wave   = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int lane       = threadIdx.x & 63;
  for (int64_t k = wave; k < rv.n_stored; k += nwaves) {
    uint32_t const b = (uint32_t)rv.offsets[k], len = (uint32_t)rv.offsets[k + 1] - b;
    int32_t const v  = rv.row_of(k);
    for (uint32_t p = lane; p < len; p += 64) rows[b + p] = v;
  }
}
}  // namespace
}  // namespace cga

extern "C" cugraph_error_code_t cugraph_decompress_to_edgelist(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph, bool_t do_expensive_check,
                                                               cugraph_edgelist_t** result, cugraph_error_t** error)
{
  (void)do_expensive_check;
  if (result) *result = nullptr;
  return guarded(error, [&] {
    handle_t const& h = H(handle);
    CGA_EXPECTS(graph != nullptr && result != nullptr, CUGRAPH_INVALID_INPUT, "graph / result is NULL");
    graph_t& g = GM(graph);
    HIP_TRY(hipSetDevice(h.device));
    if (g.mg) {
      // A graph from cugraph_graph_create_mg: every rank returns ITS part of the edge list (decompress_to_edgelist.cpp:60-103 with multi_gpu =
      // true returns the rank's local partition) -- here the slice the rank holds after the creation flags, external ids, with the weights and
      // the edge ids / edge type ids that came with it.  Not collective.  The union over the ranks is the graph's edge multiset.
      mg_graph_t const& mg = *g.mg;
      edge_list_t const& el = mg.el;
      auto out = std::make_unique<edgelist_result_t>();
      auto take = [&](void const* p, size_t elem, cugraph_data_type_id_t t) {
        auto* a = new device_array_t((size_t)el.n, t);
        if (el.n > 0) HIP_TRY(hipMemcpyAsync(a->buf.ptr, p, (size_t)el.n * elem, hipMemcpyDeviceToDevice, h.stream));
        return a;
      };
      out->src = take(el.s.data(), 4, INT32);
      out->dst = take(el.d.data(), 4, INT32);
      if (g.has_weights) out->wgt = take(el.w.ptr, el.wsize, g.weight_type);
      if (g.has_edge_ids) out->ids = take(mg.edge_ids.ptr, mg.ids_size, g.edge_id_type);
      if (g.has_edge_types) out->types = take(mg.edge_types.data(), 4, INT32);
      h.sync();
      outer_replace_ids(h, g, out->src);
      outer_replace_ids(h, g, out->dst);
      *result = reinterpret_cast<cugraph_edgelist_t*>(out.release());
      return;
    }
    ensure_orientation(h, g, false, /*dcs_aware=*/true);
    auto out = std::make_unique<edgelist_result_t>();
    out->src = new device_array_t((size_t)g.ne, g.vertex_type);
    out->dst = new device_array_t((size_t)g.ne, g.vertex_type);
    if (g.ne > 0) {
      hipLaunchKernelGGL(k_rows_of_edges, grid_for(g.nv * 16, kBlock, 8192), kBlock, 0, h.stream, rows_view(g.csr, g.nv), out->src->buf.as<int32_t>());
      HIP_TRY(hipMemcpyAsync(out->dst->buf.ptr, g.csr.indices.data(), (size_t)g.ne * 4, hipMemcpyDeviceToDevice, h.stream));
      unrenumber_int_to_ext(h, g, out->src->buf.as<int32_t>(), g.ne);
      unrenumber_int_to_ext(h, g, out->dst->buf.as<int32_t>(), g.ne);
      if (g.has_weights) {
        out->wgt = new device_array_t((size_t)g.ne, g.weight_type);
        HIP_TRY(hipMemcpyAsync(out->wgt->buf.ptr, g.csr.weights.ptr, (size_t)g.ne * dtype_size(g.weight_type), hipMemcpyDeviceToDevice, h.stream));
      }
    } else if (g.has_weights) {
      out->wgt = new device_array_t(0, g.weight_type);
    }
    if (g.has_edge_ids) {
      out->ids = new device_array_t((size_t)g.ne, g.edge_id_type);
      if (g.ne > 0) HIP_TRY(hipMemcpyAsync(out->ids->buf.ptr, g.csr.edge_ids.ptr, (size_t)g.ne * dtype_size(g.edge_id_type), hipMemcpyDeviceToDevice, h.stream));
    }
    if (g.has_edge_types) {
      out->types = new device_array_t((size_t)g.ne, INT32);
      if (g.ne > 0) HIP_TRY(hipMemcpyAsync(out->types->buf.ptr, g.csr.edge_types.data(), (size_t)g.ne * 4, hipMemcpyDeviceToDevice, h.stream));
    }
    h.sync();
    outer_replace_ids(h, g, out->src);
    outer_replace_ids(h, g, out->dst);
    *result = reinterpret_cast<cugraph_edgelist_t*>(out.release());
  });
}
static cugraph_type_erased_device_array_view_t* el_view(device_array_t* a)
{
  return a ? reinterpret_cast<cugraph_type_erased_device_array_view_t*>(a->new_view()) : nullptr;
}
extern "C" cugraph_type_erased_device_array_view_t* cugraph_edgelist_get_sources(cugraph_edgelist_t* e) { return el_view(reinterpret_cast<edgelist_result_t*>(e)->src); }
extern "C" cugraph_type_erased_device_array_view_t* cugraph_edgelist_get_destinations(cugraph_edgelist_t* e) { return el_view(reinterpret_cast<edgelist_result_t*>(e)->dst); }
extern "C" cugraph_type_erased_device_array_view_t* cugraph_edgelist_get_edge_weights(cugraph_edgelist_t* e) { return el_view(reinterpret_cast<edgelist_result_t*>(e)->wgt); }
extern "C" cugraph_type_erased_device_array_view_t* cugraph_edgelist_get_edge_ids(cugraph_edgelist_t* e) { return el_view(reinterpret_cast<edgelist_result_t*>(e)->ids); }
extern "C" cugraph_type_erased_device_array_view_t* cugraph_edgelist_get_edge_type_ids(cugraph_edgelist_t* e) { return el_view(reinterpret_cast<edgelist_result_t*>(e)->types); }
extern "C" cugraph_type_erased_device_array_view_t* cugraph_edgelist_get_edge_offsets(cugraph_edgelist_t*) { return nullptr; }
extern "C" void cugraph_edgelist_free(cugraph_edgelist_t* e) { delete reinterpret_cast<edgelist_result_t*>(e); }
