// Multi-GPU graph construction behind cugraph_graph_create_mg (design: mg_graph.hpp).  Replaces cpp/src/c_api/graph_mg.cpp:140-560,
// the multi_gpu halves of cpp/src/structure/create_graph_from_edgelist_impl.cuh:473-955 and renumber_edgelist_impl.cuh:425-829, and
// the edge shuffle of cpp/include/cugraph/utilities/shuffle_comm.cuh:139-186 -- on the library's own communicator (comm.hpp).
#include "mg_graph.hpp"

#include "comm.hpp"

#include <algorithm>

namespace cga {

namespace {

int bits_for(uint64_t max_value)
{
  int b = 0;
  while (b < 64 && (max_value >> b) != 0) ++b;
  return b < 1 ? 1 : b;
}

// ------------------------------------------------------------------------------------------------ small kernels
__global__ void k_present(uint32_t const* a, uint32_t const* b, uint32_t const* listed, int64_t n, uint32_t* present)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) present[i] = (a[i] | b[i] | (listed ? listed[i] : 0u)) != 0u ? 1u : 0u;
}

// owner of the unordered endpoint pair of an edge (all edges between two vertices, either direction, land on one rank)
__global__ void k_pair_owner(int32_t const* s, int32_t const* d, int64_t n, int P, int32_t* owner)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t const a = (uint32_t)min(s[i], d[i]), b = (uint32_t)max(s[i], d[i]);
    uint32_t x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu;
    x ^= x >> 16;
    owner[i] = (int32_t)(x % (uint32_t)P);
  }
}

__global__ void k_mark_listed(int32_t const* v, int64_t n, int64_t vmin, uint32_t* flags)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) flags[(int64_t)v[i] - vmin] = 1u;
}

// per-bin counts for a handful of bins (owners): one ballot per bin instead of same-address atomics
__global__ void __launch_bounds__(256) k_count_bins(int32_t const* key, int64_t n, int nbins, unsigned long long* counts)
{
  int const lane = threadIdx.x & 63;
  unsigned long long mine = 0;  // lane b accumulates bin b (bins 64.. wrap around: nbins <= 64)
  int64_t const stride = (int64_t)gridDim.x * blockDim.x;
  int64_t const nround = (n + stride - 1) / stride;
  for (int64_t k = 0; k < nround; ++k) {
    int64_t const i = k * stride + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int const v     = i < n ? key[i] : -1;
    for (int b = 0; b < nbins; ++b) {
      unsigned long long const m = __ballot(v == b);
      if (lane == b) mine += (unsigned long long)__popcll(m);
    }
  }
  if (lane < nbins && mine) atomicAdd(&counts[lane], mine);
}

__global__ void k_sum_u32(uint32_t const* v, int64_t n, unsigned long long* out)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  unsigned long long acc = 0;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) acc += v[i];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

// degree order: present vertices by descending degree (ties: ascending id -- the sort is stable over ascending ids), absent ids last
__global__ void k_order_keys(uint32_t const* deg, uint32_t const* present, int64_t n, uint32_t maxdeg, uint64_t* keys, uint32_t* vals)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    keys[i] = present[i] ? (uint64_t)(maxdeg - deg[i]) : (uint64_t)maxdeg + 1u;
    vals[i] = (uint32_t)i;
  }
}

__global__ void k_positions(uint32_t const* order, int64_t n, int64_t nv_global, int32_t* pos)
{
  int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; p < n; p += (int64_t)gridDim.x * blockDim.x) pos[order[p]] = p < nv_global ? (int32_t)p : -1;
}

// what travels with an edge and where to.  mode 0 (PageRank): to the owner of the DESTINATION: a = position of the source,
// b = local row of the destination.  mode 1 (traversal, out-edges): to the owner of the SOURCE: a = local row of the source,
// b = compact global id of the destination.  mode 2 (traversal, in-edges): to the owner of the DESTINATION: a = local row of the
// destination, b = compact global id of the source, c = external id of the source - vmin.
__global__ void k_route(int32_t const* s, int32_t const* d, int64_t m, int64_t vmin, int32_t const* pos, int P, int64_t L, int mode,
                        int32_t* owner, int32_t* a, int32_t* b, int32_t* c)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t const ps = pos[(int64_t)s[i] - vmin], pd = pos[(int64_t)d[i] - vmin];
    if (mode == 0) {
      owner[i] = pd % P; a[i] = ps; b[i] = pd / P;
    } else if (mode == 1) {
      owner[i] = ps % P; a[i] = ps / P; b[i] = (int32_t)((int64_t)(pd % P) * L + pd / P);
    } else {
      owner[i] = pd % P; a[i] = pd / P; b[i] = (int32_t)((int64_t)(ps % P) * L + ps / P); c[i] = (int32_t)((int64_t)s[i] - vmin);
    }
  }
}

__global__ void k_owner_keys(int32_t const* owner, int64_t m, uint64_t* keys, uint32_t* vals)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < m; i += (int64_t)gridDim.x * blockDim.x) { keys[i] = (uint64_t)owner[i]; vals[i] = (uint32_t)i; }
}

// PageRank columns: mark the source positions this rank references, de-interleaved by owner: slot (p % P) * Lc + p / P
__global__ void k_mark_columns(int32_t const* ps, int64_t m, int P, int64_t Lc, uint32_t* flags)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < m; i += (int64_t)gridDim.x * blockDim.x) { int32_t const p = ps[i]; flags[(int64_t)(p % P) * Lc + p / P] = 1u; }
}
__global__ void k_edge_columns(int32_t const* ps, int64_t m, int P, int64_t Lc, uint32_t const* colrank, int32_t* col)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < m; i += (int64_t)gridDim.x * blockDim.x) { int32_t const p = ps[i]; col[i] = (int32_t)colrank[(int64_t)(p % P) * Lc + p / P]; }
}
// request list: the local rows (at their owner) of the marked columns, in column order
__global__ void k_requests(uint32_t const* flags, uint32_t const* colrank, int64_t n_slots, int64_t Lc, int32_t* req)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n_slots; i += (int64_t)gridDim.x * blockDim.x)
    if (flags[i]) req[colrank[i]] = (int32_t)(i % Lc);
}

// out[r] = table[order[r * P + rank]]: a global per-vertex table -> the owned rows
template <typename TO, typename TI>
__global__ void k_take_owned(TI const* table, uint32_t const* order, int64_t n_rows, int P, int rank, TO* out)
{
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; r < n_rows; r += (int64_t)gridDim.x * blockDim.x) out[r] = (TO)table[order[r * P + rank]];
}
__global__ void k_owned_ids(uint32_t const* order, int64_t n_rows, int P, int rank, int64_t vmin, int32_t* out)
{
  int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; r < n_rows; r += (int64_t)gridDim.x * blockDim.x) out[r] = (int32_t)((int64_t)order[r * P + rank] + vmin);
}

template <typename WT>
__global__ void k_add_weights_f64(int32_t const* s, WT const* w, int64_t m, int64_t vmin, double* acc)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < m; i += (int64_t)gridDim.x * blockDim.x) atomicAdd(&acc[(int64_t)s[i] - vmin], (double)w[i]);
}
__global__ void k_u32_to_f64(uint32_t const* in, int64_t n, double* out)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (double)in[i];
}

__global__ void k_pack_row_key(int32_t const* row, int32_t const* minor, int64_t m, uint64_t* keys, uint32_t* vals)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < m; i += (int64_t)gridDim.x * blockDim.x) { keys[i] = ((uint64_t)(uint32_t)row[i] << 32) | (uint32_t)minor[i]; vals[i] = (uint32_t)i; }
}
__global__ void k_to_float(double const* in, int64_t n, float* out)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (float)in[i];
}

// ------------------------------------------------------------------------------------------------ helpers
std::vector<int64_t> count_owners(handle_t const& h, int32_t const* owner, int64_t m, int P)
{
  dvec<unsigned long long> c(64);
  HIP_TRY(hipMemsetAsync(c.data(), 0, 64 * sizeof(unsigned long long), h.stream));
  if (m > 0) hipLaunchKernelGGL(k_count_bins, grid_for(m, 256, 2048), 256, 0, h.stream, owner, m, P, c.data());
  std::vector<unsigned long long> hc(64);
  HIP_TRY(hipMemcpyAsync(hc.data(), c.data(), 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h.stream));
  h.sync();
  std::vector<int64_t> out(P);
  for (int r = 0; r < P; ++r) out[r] = (int64_t)hc[r];
  return out;
}

using column_t = mg_column_t;

}  // namespace

// Sends every element i of the columns to rank owner[i]; returns the received columns (grouped by sender) and their length.
int64_t mg_shuffle_by_owner(handle_t const& h, comm_t& c, int32_t const* owner, int64_t m, std::vector<mg_column_t> const& cols, std::vector<dev_buf>& out)
{
  int const P = c.size;
  out.clear();
  out.resize(cols.size());
  std::vector<int64_t> counts = count_owners(h, owner, m, P), rc;
  dvec<uint64_t> keys, keys_tmp;
  dvec<uint32_t> perm, perm_tmp;
  size_t const m1 = (size_t)std::max<int64_t>(m, 1);
  keys.resize_discard(m1); keys_tmp.resize_discard(m1); perm.resize_discard(m1); perm_tmp.resize_discard(m1);
  if (m > 0) {
    hipLaunchKernelGGL(k_owner_keys, grid_for(m, kBlock, 4096), kBlock, 0, h.stream, owner, m, keys.data(), perm.data());
    radix_sort_u64_u32(h, keys.data(), perm.data(), keys_tmp.data(), perm_tmp.data(), m, 0, bits_for((uint64_t)(P - 1)));  // stable: input order inside an owner
  }
  keys = dvec<uint64_t>(); keys_tmp = dvec<uint64_t>(); perm_tmp = dvec<uint32_t>();
  int64_t received = 0;
  for (size_t k = 0; k < cols.size(); ++k) {
    dev_buf grouped(m1 * cols[k].elem);
    if (m > 0) {
      if (cols[k].elem == 4) gather_b32(h, static_cast<uint32_t const*>(cols[k].ptr), perm.data(), grouped.as<uint32_t>(), m);
      else gather_b64(h, static_cast<uint64_t const*>(cols[k].ptr), perm.data(), grouped.as<uint64_t>(), m);
    }
    c.all_to_all_v(h, grouped.ptr, counts, cols[k].elem, out[k], rc);
    received = 0;
    for (auto x : rc) received += x;
  }
  return received;
}

namespace {
int64_t shuffle_by_owner(handle_t const& h, comm_t& c, int32_t const* owner, int64_t m, std::vector<mg_column_t> const& cols, std::vector<dev_buf>& out)
{
  return mg_shuffle_by_owner(h, c, owner, m, cols, out);
}

struct vertex_order_t {
  dvec<uint32_t> order;  // position -> external id - vmin (positions >= nv_global: ids that are no vertices)
  dvec<int32_t> pos;     // external id - vmin -> position (-1: not a vertex)
};

// the global degree order every rank computes for itself from the all-reduced degrees (identical inputs, deterministic sort)
void degree_order(handle_t const& h, uint32_t const* deg, uint32_t const* present, int64_t vrange, int64_t nv_global, vertex_order_t& vo)
{
  int32_t mn = 0, mx = 0;
  minmax_i32(h, reinterpret_cast<int32_t const*>(deg), vrange, &mn, &mx);  // degrees are below 2^31
  dvec<uint64_t> keys((size_t)vrange), keys_tmp((size_t)vrange);
  dvec<uint32_t> vals_tmp((size_t)vrange);
  vo.order.resize_discard((size_t)vrange);
  vo.pos.resize_discard((size_t)vrange);
  hipLaunchKernelGGL(k_order_keys, grid_for(vrange, kBlock, 4096), kBlock, 0, h.stream, deg, present, vrange, (uint32_t)mx, keys.data(), vo.order.data());
  radix_sort_u64_u32(h, keys.data(), vo.order.data(), keys_tmp.data(), vals_tmp.data(), vrange, 0, bits_for((uint64_t)mx + 1));
  hipLaunchKernelGGL(k_positions, grid_for(vrange, kBlock, 4096), kBlock, 0, h.stream, (uint32_t const*)vo.order.data(), vrange, nv_global, vo.pos.data());
  h.sync();
}

// global (all-reduced) per-id counts of one endpoint column of this rank's slice
void global_degree(handle_t const& h, comm_t& c, mg_graph_t const& mg, int32_t const* ids, dvec<uint32_t>& out)
{
  out.resize_discard((size_t)mg.vrange);
  HIP_TRY(hipMemsetAsync(out.data(), 0, (size_t)mg.vrange * 4, h.stream));
  if (mg.el.n > 0) histogram_i32_mapped(h, ids, mg.el.n, mg.vmin, nullptr, out.data(), mg.vrange);
  c.all_reduce_sum_u32(h, out.data(), mg.vrange);
}

}  // namespace

mg_pagerank_part_t::~mg_pagerank_part_t()
{
  if (local) cugraph_graph_free(local);
}

// ------------------------------------------------------------------------------------------------ graph creation
namespace {
__global__ void k_scatter_paths(int32_t const* v, int32_t const* dist, int32_t const* pred, int64_t n, int64_t vmin, int64_t vrange, uint32_t* dist1, uint32_t* pred2)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t const k = (int64_t)v[i] - vmin;
    if (k < 0 || k >= vrange) continue;
    dist1[k] = (uint32_t)dist[i] + 1u;                 // hop counts are 0 .. INT32_MAX (unreached)
    pred2[k] = (uint32_t)(pred[i] < 0 ? -1 : pred[i]) + 2u;  // -1 = no predecessor
  }
}

// every rank's `mine` elements of `elem` bytes (a multiple of 4), concatenated in rank order, on every rank (collective)
int64_t all_gather_v(handle_t const& h, comm_t& c, void const* in, int64_t mine, size_t elem, dev_buf& all)
{
  int const P = c.size;
  std::vector<int64_t> counts(P);
  c.host_allgather(&mine, sizeof(mine), counts.data());
  int64_t stride = 1, total = 0;
  for (auto x : counts) { stride = std::max(stride, x); total += x; }
  dev_buf padded_in((size_t)stride * elem), padded((size_t)stride * elem * P);
  HIP_TRY(hipMemsetAsync(padded_in.ptr, 0, (size_t)stride * elem, h.stream));
  if (mine > 0) HIP_TRY(hipMemcpyAsync(padded_in.ptr, in, (size_t)mine * elem, hipMemcpyDeviceToDevice, h.stream));
  c.all_gather(h, padded_in.ptr, (size_t)stride * elem, padded.ptr);
  all.alloc((size_t)std::max<int64_t>(total, 1) * elem);
  int64_t at = 0;
  for (int r = 0; r < P; ++r) {
    if (counts[r] > 0)
      HIP_TRY(hipMemcpyAsync(static_cast<char*>(all.ptr) + (size_t)at * elem, static_cast<char const*>(padded.ptr) + (size_t)r * stride * elem, (size_t)counts[r] * elem,
                             hipMemcpyDeviceToDevice, h.stream));
    at += counts[r];
  }
  h.sync();
  return total;
}
}  // namespace

bool mg_outer_ids(handle_t const& h, device_array_view_t const* vertices, device_array_view_t const* src, device_array_view_t const* dst, outer_ids_t& outer)
{
  comm_t* cp = handle_comm(h);
  CGA_EXPECTS(cp != nullptr, CUGRAPH_INVALID_HANDLE, "multi-GPU graph: the handle carries no communicator");
  comm_t& c = *cp;
  device_array_view_t const* cols[3] = {src, dst, vertices};
  struct info_t { int64_t wide, bad; } mine{0, 0};
  for (auto v : cols) {
    if (v == nullptr) continue;
    if (v->type == INT64) mine.wide = 1;
    else if (v->type != INT32) mine.bad = 1;
  }
  std::vector<info_t> all(c.size);
  c.host_allgather(&mine, sizeof(mine), all.data());  // (the ranks decide together: nobody is left waiting in a collective)
  int64_t wide = 0, bad = 0;
  for (auto const& i : all) { wide += i.wide; bad += i.bad; }
  CGA_EXPECTS(bad == 0, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "multi-GPU graph: vertex ids must be INT32 or INT64");
  if (wide == 0) return false;
  // a mix of INT32 and INT64 columns (on one rank or between ranks) promotes the graph to INT64 (graph_sg.cpp:745-779 does the same for one rank)
  dvec<int64_t> local;
  outer_collect(h, cols, 3, local);  // this rank's distinct ids, ascending
  dev_buf gathered;
  int64_t const total = all_gather_v(h, c, local.data(), (int64_t)local.size(), sizeof(int64_t), gathered);
  local = dvec<int64_t>();
  device_array_view_t const gv{gathered.ptr, (size_t)total, INT64};
  device_array_view_t const* one[1] = {&gv};
  outer_collect(h, one, 1, outer.ext);  // the same list on every rank
  outer.active   = true;
  outer.identity = false;
  outer.type     = INT64;
  return true;
}

int64_t mg_host_max(graph_t& g, int64_t mine)
{
  comm_t& c = *g.mg->comm;
  std::vector<int64_t> all(c.size);
  c.host_allgather(&mine, sizeof(mine), all.data());
  int64_t m = mine;
  for (auto x : all) m = std::max(m, x);
  return m;
}

void mg_agree(graph_t& g, std::function<void()> const& local_checks, char const* what)
{
  comm_t& c = *g.mg->comm;
  int32_t mine = 0;
  std::string msg;
  cugraph_error_code_t code = CUGRAPH_SUCCESS;
  try {
    local_checks();
  } catch (api_error const& e) {
    mine = 1; code = e.code; msg = e.what();
  }
  std::vector<int32_t> all(c.size);
  c.host_allgather(&mine, sizeof(mine), all.data());
  if (mine) throw api_error(code, msg);
  for (int r = 0; r < c.size; ++r)
    if (all[r]) throw api_error(CUGRAPH_INVALID_INPUT, std::string(what) + ": rank " + std::to_string(r) + " rejected its arguments (a collective call fails on every rank)");
}

void mg_agree_same(graph_t& g, void const* blob, size_t bytes, char const* what)
{
  comm_t& c = *g.mg->comm;
  CGA_EXPECTS(bytes <= 64, CUGRAPH_UNKNOWN_ERROR, "mg_agree_same: at most 64 bytes");
  unsigned char mine[64] = {0};
  std::memcpy(mine, blob, bytes);
  std::vector<unsigned char> all((size_t)c.size * 64);
  c.host_allgather(mine, 64, all.data());
  for (int r = 0; r < c.size; ++r)
    CGA_EXPECTS(std::memcmp(all.data() + (size_t)r * 64, all.data(), 64) == 0, CUGRAPH_INVALID_INPUT,
                std::string(what) + ": the ranks pass different scalar arguments (rank " + std::to_string(r) + " differs from rank 0)");
}

void mg_gather_paths(handle_t const& h, graph_t& g, int32_t const* vertices, int32_t const* dist, int32_t const* pred, int64_t n, dvec<uint32_t>& dist1,
                     dvec<uint32_t>& pred2)
{
  mg_graph_t& mg = *g.mg;
  comm_t& c      = *mg.comm;
  dist1.resize_discard((size_t)mg.vrange);
  pred2.resize_discard((size_t)mg.vrange);
  HIP_TRY(hipMemsetAsync(dist1.data(), 0, (size_t)mg.vrange * 4, h.stream));
  HIP_TRY(hipMemsetAsync(pred2.data(), 0, (size_t)mg.vrange * 4, h.stream));
  if (n > 0) hipLaunchKernelGGL(k_scatter_paths, grid_for(n, kBlock, 8192), kBlock, 0, h.stream, vertices, dist, pred, n, mg.vmin, mg.vrange, dist1.data(), pred2.data());
  // every id is reported by exactly one rank (its owner), the others contribute 0: the sum IS the gather
  c.all_reduce_sum_u32(h, dist1.data(), mg.vrange);
  c.all_reduce_sum_u32(h, pred2.data(), mg.vrange);
}

void mg_graph_create(handle_t const& h, graph_t& g, device_array_view_t const* vertices, device_array_view_t const* src, device_array_view_t const* dst,
                     device_array_view_t const* weights, device_array_view_t const* edge_ids, device_array_view_t const* edge_type_ids, bool drop_self_loops,
                     bool drop_multi_edges, bool symmetrize)
{
  comm_t* cp = handle_comm(h);
  CGA_EXPECTS(cp != nullptr, CUGRAPH_INVALID_HANDLE, "multi-GPU graph: the handle carries no communicator");
  comm_t& c = *cp;
  CGA_EXPECTS(src->type == INT32 && dst->type == INT32 && (vertices == nullptr || vertices->type == INT32), CUGRAPH_UNSUPPORTED_TYPE_COMBINATION,
              "multi-GPU graphs take INT32 vertex ids in this build");
  CGA_EXPECTS(weights == nullptr || weights->type == FLOAT32 || weights->type == FLOAT64, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "weights must be FLOAT32 or FLOAT64");
  auto mg  = std::make_shared<mg_graph_t>();
  mg->comm = cp;
  int64_t const m = (int64_t)src->size;
  // Edge ids / edge type ids: edge PROPERTIES that the sampling and lookup families read (graph_mg.cpp:127-151 shuffles them with their edges);
  // none of the algorithms of this library does.  They stay with this rank's slice and come back from cugraph_decompress_to_edgelist.  As
  // on one GPU (graph.hip: create_sg) they are refused together with a flag that rewrites the edge list.
  int64_t const props_state = [&]() -> int64_t {
    if (edge_ids == nullptr && edge_type_ids == nullptr) return 0;
    if (drop_self_loops || drop_multi_edges || symmetrize) return -1;
    if ((edge_ids && edge_ids->size != src->size) || (edge_type_ids && edge_type_ids->size != src->size)) return -2;
    if ((edge_ids && edge_ids->type != INT32 && edge_ids->type != INT64) || (edge_type_ids && edge_type_ids->type != INT32)) return -3;
    return (edge_ids ? (edge_ids->type == INT64 ? 2 : 1) : 0) + (edge_type_ids ? 4 : 0);
  }();
  CGA_EXPECTS(m <= kMaxSignedEdges, CUGRAPH_INVALID_INPUT, "multi-GPU graph: a rank's slice must hold fewer than 2^31 edges");
  edge_list_t& el = mg->el;
  el.n     = m;
  el.wsize = weights ? dtype_size(weights->type) : 0;
  el.s.resize_discard((size_t)std::max<int64_t>(m, 1));
  el.d.resize_discard((size_t)std::max<int64_t>(m, 1));
  if (m > 0) {
    HIP_TRY(hipMemcpyAsync(el.s.data(), src->data, (size_t)m * 4, hipMemcpyDeviceToDevice, h.stream));
    HIP_TRY(hipMemcpyAsync(el.d.data(), dst->data, (size_t)m * 4, hipMemcpyDeviceToDevice, h.stream));
    if (weights) {
      el.w.alloc((size_t)m * el.wsize);
      HIP_TRY(hipMemcpyAsync(el.w.ptr, weights->data, (size_t)m * el.wsize, hipMemcpyDeviceToDevice, h.stream));
    }
  }
  h.sync();
  if (drop_self_loops && el.n > 0) edgelist_drop_self_loops(h, el);
  mg->n_listed = vertices ? (int64_t)vertices->size : 0;
  if (mg->n_listed > 0) {
    mg->listed.resize_discard((size_t)mg->n_listed);
    HIP_TRY(hipMemcpyAsync(mg->listed.data(), vertices->data, (size_t)mg->n_listed * 4, hipMemcpyDeviceToDevice, h.stream));
    h.sync();
  }
  // the dense id range over all ranks, edge total, does anybody list vertices, do all ranks agree on the weights
  struct info_t { int64_t lo, hi, ne, listed, weighted, wsize, props; } mine{INT64_MAX, INT64_MIN, el.n, mg->n_listed, weights ? 1 : 0, (int64_t)el.wsize, props_state};
  auto upd = [&](int32_t const* p, int64_t n) {
    if (n <= 0) return;
    int32_t a, b;
    minmax_i32(h, p, n, &a, &b);
    mine.lo = std::min<int64_t>(mine.lo, a); mine.hi = std::max<int64_t>(mine.hi, b);
  };
  upd(el.s.data(), el.n); upd(el.d.data(), el.n);
  if (mg->n_listed > 0) upd(mg->listed.data(), mg->n_listed);
  std::vector<info_t> all(c.size);
  c.host_allgather(&mine, sizeof(mine), all.data());
  int64_t lo = INT64_MAX, hi = INT64_MIN, ne = 0, any_listed = 0;
  for (auto const& i : all) {
    lo = std::min(lo, i.lo); hi = std::max(hi, i.hi); ne += i.ne; any_listed += i.listed;
    CGA_EXPECTS(i.weighted == mine.weighted && i.wsize == mine.wsize, CUGRAPH_INVALID_INPUT, "multi-GPU graph: the ranks disagree on the edge weights (present / type)");
  }
  // (decided on the gathered states, so that every rank throws or none does)
  for (auto const& i : all) {
    CGA_EXPECTS(i.props != -1, CUGRAPH_NOT_IMPLEMENTED,
                "edge ids / edge type ids together with drop_self_loops / drop_multi_edges / symmetrize are not supported: the rewritten edge list has no one-to-one relation to them");
    CGA_EXPECTS(i.props != -2, CUGRAPH_INVALID_INPUT, "Invalid input arguments: src size != edge id / edge type prop size");
    CGA_EXPECTS(i.props != -3, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "edge ids must be INT32 or INT64, edge type ids INT32");
    CGA_EXPECTS(i.props == mine.props, CUGRAPH_INVALID_INPUT, "multi-GPU graph: the ranks disagree on the edge ids / edge type ids (present / type)");
  }
  if (edge_ids) {
    mg->ids_size = dtype_size(edge_ids->type);
    mg->edge_ids.alloc((size_t)std::max<int64_t>(m, 1) * mg->ids_size);
    if (m > 0) HIP_TRY(hipMemcpyAsync(mg->edge_ids.ptr, edge_ids->data, (size_t)m * mg->ids_size, hipMemcpyDeviceToDevice, h.stream));
    g.has_edge_ids = true;
    g.edge_id_type = edge_ids->type;
  }
  if (edge_type_ids) {
    mg->has_edge_types = true;
    mg->edge_types.resize_discard((size_t)std::max<int64_t>(m, 1));
    if (m > 0) HIP_TRY(hipMemcpyAsync(mg->edge_types.data(), edge_type_ids->data, (size_t)m * 4, hipMemcpyDeviceToDevice, h.stream));
    g.has_edge_types = true;
  }
  h.sync();
  CGA_EXPECTS(hi >= lo, CUGRAPH_INVALID_INPUT, "multi-GPU graph: no edges and no vertices on any rank");
  CGA_EXPECTS(hi - lo + 1 < ((int64_t)1 << 31) - 2, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "multi-GPU graph: the external id range must fit 31 bits");
  mg->vmin = lo; mg->vrange = hi - lo + 1; mg->ne_global = ne;
  if ((drop_multi_edges || symmetrize) && ne > 0) {
    // graph_mg.cpp:165-230 (remove_multi_edges / symmetrize_edgelist on the shuffled list).  Both decisions concern the edges between ONE
    // unordered pair of endpoints: those are brought together on one rank (owner = a hash of the pair), where the single-GPU routines
    // (edgelist.hip) decide as they would on the whole list.  Which rank holds an edge afterwards does not matter: the partitions
    // re-shuffle the slices on first use.
    size_t const m1 = (size_t)std::max<int64_t>(el.n, 1);
    dvec<int32_t> owner(m1);
    if (el.n > 0) hipLaunchKernelGGL(k_pair_owner, grid_for(el.n, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)el.s.data(), (int32_t const*)el.d.data(), el.n, c.size, owner.data());
    std::vector<mg_column_t> cols{{el.s.data(), 4}, {el.d.data(), 4}};
    if (el.wsize) cols.push_back({el.w.ptr, el.wsize});
    std::vector<dev_buf> got;
    int64_t const n_in = mg_shuffle_by_owner(h, c, owner.data(), el.n, cols, got);
    CGA_EXPECTS(n_in <= kMaxSignedEdges, CUGRAPH_INVALID_INPUT, "multi-GPU graph: a rank's share of the edge pairs must hold fewer than 2^31 edges");
    size_t const n1 = (size_t)std::max<int64_t>(n_in, 1);
    el.n = n_in;
    el.s.resize_discard(n1); el.d.resize_discard(n1);
    if (n_in > 0) {
      HIP_TRY(hipMemcpyAsync(el.s.data(), got[0].ptr, (size_t)n_in * 4, hipMemcpyDeviceToDevice, h.stream));
      HIP_TRY(hipMemcpyAsync(el.d.data(), got[1].ptr, (size_t)n_in * 4, hipMemcpyDeviceToDevice, h.stream));
      if (el.wsize) {
        el.w.alloc((size_t)n_in * el.wsize);
        HIP_TRY(hipMemcpyAsync(el.w.ptr, got[2].ptr, (size_t)n_in * el.wsize, hipMemcpyDeviceToDevice, h.stream));
      }
    }
    h.sync();
    if (el.n > 0 && drop_multi_edges) edgelist_drop_multi_edges(h, el, mg->vmin, mg->vrange);
    if (el.n > 0 && symmetrize) edgelist_symmetrize(h, el, mg->vmin, mg->vrange);
    std::vector<int64_t> counts(c.size);
    int64_t const mine_n = el.n;
    c.host_allgather(&mine_n, sizeof(mine_n), counts.data());
    ne = 0;
    for (auto x : counts) ne += x;
    mg->ne_global = ne;
  }
  // which ids are vertices: an endpoint of an edge anywhere, or listed anywhere
  dvec<uint32_t> din, dout, lst;
  global_degree(h, c, *mg, el.d.data(), din);
  global_degree(h, c, *mg, el.s.data(), dout);
  if (any_listed > 0) {
    lst.resize_discard((size_t)mg->vrange);
    HIP_TRY(hipMemsetAsync(lst.data(), 0, (size_t)mg->vrange * 4, h.stream));
    if (mg->n_listed > 0) hipLaunchKernelGGL(k_mark_listed, grid_for(mg->n_listed, kBlock, 4096), kBlock, 0, h.stream, (int32_t const*)mg->listed.data(), mg->n_listed, mg->vmin, lst.data());
    c.all_reduce_sum_u32(h, lst.data(), mg->vrange);
  }
  mg->present.resize_discard((size_t)mg->vrange);
  hipLaunchKernelGGL(k_present, grid_for(mg->vrange, kBlock, 4096), kBlock, 0, h.stream, (uint32_t const*)din.data(), (uint32_t const*)dout.data(),
                     any_listed > 0 ? (uint32_t const*)lst.data() : (uint32_t const*)nullptr, mg->vrange, mg->present.data());
  dvec<unsigned long long> cnt(1);
  HIP_TRY(hipMemsetAsync(cnt.data(), 0, 8, h.stream));
  hipLaunchKernelGGL(k_sum_u32, grid_for(mg->vrange, kBlock, 1024), kBlock, 0, h.stream, (uint32_t const*)mg->present.data(), mg->vrange, cnt.data());
  unsigned long long nvg = 0;
  h.read_back(&nvg, cnt.data(), 1);
  mg->nv_global = (int64_t)nvg;
  g.nv = mg->nv_global;
  g.ne = mg->ne_global;
  g.mg = mg;
}

namespace {
__global__ void k_mg_has_vertex(int32_t const* v, int64_t n, int64_t vmin, int64_t vrange, uint32_t const* present, uint8_t* out)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t const k = (int64_t)v[i] - vmin;
    out[i] = (k >= 0 && k < vrange && present[k]) ? 1 : 0;
  }
}
}  // namespace

namespace {
// flag[i] = 1 when ids[i] (or, ids == nullptr, the id vmin + i) is a vertex this rank answers for: (id - vmin) % P == rank; *bad counts listed ids that are no vertices
__global__ void k_mg_mine(int32_t const* ids, int64_t n, int64_t vmin, int64_t vrange, uint32_t const* present, int P, int rank, uint32_t* flag, unsigned long long* bad)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t const k = ids ? (int64_t)ids[i] - vmin : i;
    bool const ok   = k >= 0 && k < vrange && present[k] != 0u;
    if (ids && !ok) atomicAdd(bad, 1ull);
    flag[i] = ok && (int)(k % P) == rank ? 1u : 0u;
  }
}
__global__ void k_mg_take(int32_t const* ids, uint32_t const* flag, uint32_t const* pos, int64_t n, int64_t vmin, uint32_t const* din, uint32_t const* dout, int32_t* out_ids,
                          int32_t* out_in, int32_t* out_out)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (flag[i]) {
      int64_t const k  = ids ? (int64_t)ids[i] - vmin : i;
      uint32_t const q = pos[i];
      out_ids[q] = (int32_t)(k + vmin);
      if (out_in) out_in[q] = (int32_t)din[k];
      if (out_out) out_out[q] = (int32_t)dout[k];
    }
}
}  // namespace

// cugraph_degrees / _in_degrees / _out_degrees on a multi-GPU graph (degrees.cu:24-213 with multi_gpu = true): collective.  Every rank
// answers for the vertices with (id - vmin) % P == rank -- all of them, or those of them ANY rank listed (the reference shuffles the listed
// vertices to their owners).  Degrees = all-reduced endpoint counts of the ranks' slices.
int64_t mg_degrees(handle_t const& h, graph_t& g, device_array_view_t const* listed, bool want_in, bool want_out, dvec<int32_t>& ids, dvec<int32_t>& in_deg, dvec<int32_t>& out_deg)
{
  mg_graph_t& mg = *g.mg;
  comm_t& c      = *mg.comm;
  int const P    = c.size;
  dvec<uint32_t> din, dout;
  if (want_in) global_degree(h, c, mg, mg.el.d.data(), din);
  if (want_out) global_degree(h, c, mg, mg.el.s.data(), dout);
  dvec<int32_t> all;
  int64_t n = mg.vrange;
  bool const any_list = [&] {
    int64_t const has = listed != nullptr ? 1 : 0;
    std::vector<int64_t> v(P);
    c.host_allgather(&has, sizeof(has), v.data());
    int64_t s = 0;
    for (auto x : v) s += x;
    return s > 0;
  }();
  if (any_list) {  // every rank's list, padded to the longest, in rank order
    int64_t const mine = listed ? (int64_t)listed->size : 0;
    std::vector<int64_t> counts(P);
    c.host_allgather(&mine, sizeof(mine), counts.data());
    int64_t stride = 1;
    for (auto x : counts) stride = std::max(stride, x);
    dvec<int32_t> in((size_t)stride);
    fill_i32(h, in.data(), stride, INT32_MIN);  // padding: outside every id range, never "mine"
    if (mine > 0) HIP_TRY(hipMemcpyAsync(in.data(), listed->data, (size_t)mine * 4, hipMemcpyDeviceToDevice, h.stream));
    dvec<int32_t> padded((size_t)stride * P);
    c.all_gather(h, in.data(), (size_t)stride * 4, padded.data());
    int64_t total = 0;
    for (auto x : counts) total += x;
    all.resize_discard((size_t)std::max<int64_t>(total, 1));
    int64_t at = 0;
    for (int r = 0; r < P; ++r) {
      if (counts[r] > 0) HIP_TRY(hipMemcpyAsync(all.data() + at, padded.data() + (size_t)r * stride, (size_t)counts[r] * 4, hipMemcpyDeviceToDevice, h.stream));
      at += counts[r];
    }
    h.sync();
    n = total;
  }
  size_t const n1 = (size_t)std::max<int64_t>(n, 1);
  dvec<uint32_t> flag(n1 + 1), pos(n1 + 1);
  dvec<unsigned long long> bad(1);
  HIP_TRY(hipMemsetAsync(bad.data(), 0, 8, h.stream));
  HIP_TRY(hipMemsetAsync(flag.data(), 0, (n1 + 1) * 4, h.stream));
  if (n > 0)
    hipLaunchKernelGGL(k_mg_mine, grid_for(n, kBlock, 8192), kBlock, 0, h.stream, any_list ? (int32_t const*)all.data() : (int32_t const*)nullptr, n, mg.vmin, mg.vrange,
                       (uint32_t const*)mg.present.data(), P, c.rank, flag.data(), bad.data());
  exclusive_scan_u32(h, flag.data(), pos.data(), n + 1);
  uint32_t n_mine = 0;
  unsigned long long n_bad = 0;
  h.read_back(&n_mine, pos.data() + n, 1);
  h.read_back(&n_bad, bad.data(), 1);
  CGA_EXPECTS(n_bad == 0, CUGRAPH_INVALID_INPUT, "Invalid input argument: source_vertices contains a vertex that is not in the graph");
  size_t const m1 = (size_t)std::max<uint32_t>(n_mine, 1);
  ids.resize_discard(m1);
  if (want_in) in_deg.resize_discard(m1);
  if (want_out) out_deg.resize_discard(m1);
  if (n > 0)
    hipLaunchKernelGGL(k_mg_take, grid_for(n, kBlock, 8192), kBlock, 0, h.stream, any_list ? (int32_t const*)all.data() : (int32_t const*)nullptr, (uint32_t const*)flag.data(),
                       (uint32_t const*)pos.data(), n, mg.vmin, want_in ? (uint32_t const*)din.data() : nullptr, want_out ? (uint32_t const*)dout.data() : nullptr, ids.data(),
                       want_in ? in_deg.data() : nullptr, want_out ? out_deg.data() : nullptr);
  h.sync();
  return (int64_t)n_mine;
}

// cugraph_has_vertex on a multi-GPU graph: is the id a vertex ANYWHERE (every rank holds the presence table)
void mg_has_vertex(handle_t const& h, graph_t const& g, int32_t const* v, int64_t n, uint8_t* out)
{
  mg_graph_t const& mg = *g.mg;
  if (n > 0) hipLaunchKernelGGL(k_mg_has_vertex, grid_for(n, kBlock, 4096), kBlock, 0, h.stream, v, n, mg.vmin, mg.vrange, (uint32_t const*)mg.present.data(), out);
}

// ------------------------------------------------------------------------------------------------ PageRank partition
mg_pagerank_part_t& mg_pagerank_part(handle_t const& h, graph_t& g)
{
  mg_graph_t& mg = *g.mg;
  if (mg.pr) return *mg.pr;
  comm_t& c       = *mg.comm;
  int const P     = c.size, me = c.rank;
  auto part       = std::make_unique<mg_pagerank_part_t>();
  part->P         = P;
  part->rank      = me;
  part->nv_global = mg.nv_global;
  build_trace tr(h, "mg pagerank");
  // global in-degrees (the schedule) and out-weight sums (PageRank's divisor)
  dvec<uint32_t> din;
  global_degree(h, c, mg, mg.el.d.data(), din);
  dvec<double> outw((size_t)mg.vrange);
  if (mg.el.wsize == 0) {
    dvec<uint32_t> dout;
    global_degree(h, c, mg, mg.el.s.data(), dout);
    hipLaunchKernelGGL(k_u32_to_f64, grid_for(mg.vrange, kBlock, 4096), kBlock, 0, h.stream, (uint32_t const*)dout.data(), mg.vrange, outw.data());
    h.sync();
  } else {
    HIP_TRY(hipMemsetAsync(outw.data(), 0, (size_t)mg.vrange * 8, h.stream));
    if (mg.el.n > 0) {
      if (mg.el.wsize == 4) hipLaunchKernelGGL(k_add_weights_f64<float>, grid_for(mg.el.n, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)mg.el.s.data(), mg.el.w.as<float const>(), mg.el.n, mg.vmin, outw.data());
      else hipLaunchKernelGGL(k_add_weights_f64<double>, grid_for(mg.el.n, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)mg.el.s.data(), mg.el.w.as<double const>(), mg.el.n, mg.vmin, outw.data());
    }
    c.all_reduce_sum_f64(h, outw.data(), mg.vrange);
  }
  tr.step("degrees");
  vertex_order_t vo;
  degree_order(h, din.data(), mg.present.data(), mg.vrange, mg.nv_global, vo);
  din = dvec<uint32_t>();
  tr.step("degree order");
  int64_t const nvg = mg.nv_global;
  part->n_rows      = nvg > me ? (nvg - me + P - 1) / P : 0;
  // owned vertices and their out-weight sums
  size_t const nr1 = (size_t)std::max<int64_t>(part->n_rows, 1);
  part->local_vertices.resize_discard(nr1);
  bool const f64 = mg.el.wsize == 8;
  part->outw_local.alloc(nr1 * (f64 ? 8 : 4));
  if (part->n_rows > 0) {
    hipLaunchKernelGGL(k_owned_ids, grid_for(part->n_rows, kBlock, 4096), kBlock, 0, h.stream, (uint32_t const*)vo.order.data(), part->n_rows, P, me, mg.vmin, part->local_vertices.data());
    if (f64) hipLaunchKernelGGL((k_take_owned<double, double>), grid_for(part->n_rows, kBlock, 4096), kBlock, 0, h.stream, (double const*)outw.data(), (uint32_t const*)vo.order.data(), part->n_rows, P, me, part->outw_local.as<double>());
    else hipLaunchKernelGGL((k_take_owned<float, double>), grid_for(part->n_rows, kBlock, 4096), kBlock, 0, h.stream, (double const*)outw.data(), (uint32_t const*)vo.order.data(), part->n_rows, P, me, part->outw_local.as<float>());
  }
  h.sync();
  outw = dvec<double>();
  // route every edge to the owner of its destination
  int64_t const m = mg.el.n;
  size_t const m1 = (size_t)std::max<int64_t>(m, 1);
  std::vector<dev_buf> got;
  int64_t e_loc = 0;
  {
    dvec<int32_t> owner(m1), a(m1), b(m1);
    if (m > 0) hipLaunchKernelGGL(k_route, grid_for(m, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)mg.el.s.data(), (int32_t const*)mg.el.d.data(), m, mg.vmin, (int32_t const*)vo.pos.data(), P, (int64_t)0, 0, owner.data(), a.data(), b.data(), (int32_t*)nullptr);
    std::vector<column_t> cols{{a.data(), 4}, {b.data(), 4}};
    if (mg.el.wsize) cols.push_back({mg.el.w.ptr, mg.el.wsize});
    e_loc = shuffle_by_owner(h, c, owner.data(), m, cols, got);
  }
  tr.step("edge shuffle");
  CGA_EXPECTS(e_loc <= kMaxSignedEdges, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "multi-GPU PageRank: a rank's share must hold fewer than 2^31 edges");
  part->ne_local      = e_loc;
  int32_t const* ps   = got[0].as<int32_t const>();
  int32_t const* rows = got[1].as<int32_t const>();
  // columns = the source positions this rank references, numbered by (owner, position): the receive window of the x exchange
  int64_t const Lc      = std::max<int64_t>((nvg + P - 1) / P, 1);
  int64_t const n_slots = (int64_t)P * Lc;
  dvec<uint32_t> flags((size_t)n_slots + 1), colrank((size_t)n_slots + 1);
  HIP_TRY(hipMemsetAsync(flags.data(), 0, ((size_t)n_slots + 1) * 4, h.stream));
  if (e_loc > 0) hipLaunchKernelGGL(k_mark_columns, grid_for(e_loc, kBlock, 8192), kBlock, 0, h.stream, ps, e_loc, P, Lc, flags.data());
  exclusive_scan_u32(h, flags.data(), colrank.data(), n_slots + 1);
  std::vector<uint32_t> seg(P + 1);
  for (int s = 0; s <= P; ++s) HIP_TRY(hipMemcpyAsync(&seg[s], colrank.data() + (int64_t)s * Lc, 4, hipMemcpyDeviceToHost, h.stream));
  h.sync();
  part->seg_start.assign(P + 1, 0);
  for (int s = 0; s <= P; ++s) part->seg_start[s] = (int64_t)seg[s];
  part->ncols = part->seg_start[P];
  size_t const e1 = (size_t)std::max<int64_t>(e_loc, 1);
  dvec<int32_t> col(e1);
  if (e_loc > 0) hipLaunchKernelGGL(k_edge_columns, grid_for(e_loc, kBlock, 8192), kBlock, 0, h.stream, ps, e_loc, P, Lc, (uint32_t const*)colrank.data(), col.data());
  dvec<int32_t> req((size_t)std::max<int64_t>(part->ncols, 1));
  hipLaunchKernelGGL(k_requests, grid_for(n_slots, kBlock, 8192), kBlock, 0, h.stream, (uint32_t const*)flags.data(), (uint32_t const*)colrank.data(), n_slots, Lc, req.data());
  h.sync();
  flags = dvec<uint32_t>(); colrank = dvec<uint32_t>();
  // tell every owner which of its rows this rank needs (in the order of this rank's window), and learn where the peers' windows want ours
  {
    std::vector<int64_t> need_counts(P), rc;
    for (int s = 0; s < P; ++s) need_counts[s] = part->seg_start[s + 1] - part->seg_start[s];
    dev_buf sidx;
    c.all_to_all_v(h, req.data(), need_counts, 4, sidx, rc);
    part->send_first.assign(P + 1, 0);
    for (int r = 0; r < P; ++r) part->send_first[r + 1] = part->send_first[r] + rc[r];
    part->n_send = part->send_first[P];
    part->send_index.resize_discard((size_t)std::max<int64_t>(part->n_send, 1));
    if (part->n_send > 0) HIP_TRY(hipMemcpyAsync(part->send_index.data(), sidx.ptr, (size_t)part->n_send * 4, hipMemcpyDeviceToDevice, h.stream));
    h.sync();
    std::vector<int64_t> mine(P + 1), all((size_t)c.size * (P + 1));
    for (int s = 0; s <= P; ++s) mine[s] = part->seg_start[s];
    c.host_allgather(mine.data(), (size_t)(P + 1) * sizeof(int64_t), all.data());
    part->dst_off.assign(P, 0);
    for (int r = 0; r < P; ++r) part->dst_off[r] = all[(size_t)r * (P + 1) + me];
  }
  tr.step("exchange plan");
  // the local CSC: rows [0, n_rows), columns = window positions [0, ncols)
  {
    int64_t const nverts = std::max<int64_t>({part->ncols, part->n_rows, 1});
    dvec<int32_t> verts((size_t)nverts);
    iota_i32(h, verts.data(), nverts, 0);
    h.sync();
    device_array_view_t vv{verts.data(), (size_t)nverts, INT32}, sv{col.data(), (size_t)e_loc, INT32}, dv{const_cast<int32_t*>(rows), (size_t)e_loc, INT32};
    device_array_view_t wv{mg.el.wsize ? got[2].ptr : nullptr, (size_t)e_loc, f64 ? FLOAT64 : FLOAT32};
    cugraph_graph_properties_t props{FALSE, TRUE};
    cugraph_error_t* err = nullptr;
    cugraph_resource_handle_t const* hh = reinterpret_cast<cugraph_resource_handle_t const*>(&h);
    // the local graph lives on a one-rank view of this handle (cugraph_graph_create_sg does not look at the communicator)
    cugraph_error_code_t const rc = cugraph_graph_create_sg(hh, &props, reinterpret_cast<cugraph_type_erased_device_array_view_t const*>(&vv),
                                                            reinterpret_cast<cugraph_type_erased_device_array_view_t const*>(&sv),
                                                            reinterpret_cast<cugraph_type_erased_device_array_view_t const*>(&dv),
                                                            mg.el.wsize ? reinterpret_cast<cugraph_type_erased_device_array_view_t const*>(&wv) : nullptr, nullptr, nullptr, TRUE, FALSE, FALSE,
                                                            FALSE, FALSE, FALSE, &part->local, &err);
    if (rc != CUGRAPH_SUCCESS) {
      std::string msg = err ? cugraph_error_message(err) : "?";
      cugraph_error_free(err);
      throw api_error(rc, "multi-GPU PageRank: building the local graph failed: " + msg);
    }
  }
  tr.step("local graph");
  mg.pr = std::move(part);
  return *mg.pr;
}

// ------------------------------------------------------------------------------------------------ 2-D PageRank partition
namespace {
// edge (s, d) -> the rank of its block, its local column and its local row (mg_pagerank2d_part_t)
__global__ void k_route2d(int32_t const* s, int32_t const* d, int64_t m, int64_t vmin, int32_t const* pos, int P, int R, int64_t L, int32_t* owner, int32_t* col, int32_t* row)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t const ps = pos[(int64_t)s[i] - vmin], pd = pos[(int64_t)d[i] - vmin];
    int const qs = ps % P, qd = pd % P;
    owner[i] = (qs / R) * R + qd % R;
    col[i]   = (int32_t)((int64_t)(qs % R) * L + ps / P);
    row[i]   = (int32_t)((int64_t)(qd / R) * L + pd / P);
  }
}
}  // namespace

void mg_grid_shape(int P, int* R, int* C)
{
  int r = 1;
  while ((r + 1) * (r + 1) <= P) ++r;
  while (P % r) --r;
  *R = r; *C = P / r;
}

mg_pagerank2d_part_t::~mg_pagerank2d_part_t()
{
  if (local) cugraph_graph_free(local);
}

mg_pagerank2d_part_t& mg_pagerank2d_part(handle_t const& h, graph_t& g)
{
  mg_graph_t& mg = *g.mg;
  if (mg.pr2d) return *mg.pr2d;
  comm_t& c   = *mg.comm;
  int const P = c.size, me = c.rank;
  auto part   = std::make_unique<mg_pagerank2d_part_t>();
  part->P = P; part->rank = me; part->nv_global = mg.nv_global;
  mg_grid_shape(P, &part->R, &part->C);
  part->r = me % part->R; part->c = me / part->R;
  int const R = part->R, C = part->C;
  build_trace tr(h, "mg pagerank 2d");
  // global in-degrees (the vertex order) and out-weight sums (PageRank's divisor): as for the 1-D partition
  dvec<uint32_t> din;
  global_degree(h, c, mg, mg.el.d.data(), din);
  dvec<double> outw((size_t)mg.vrange);
  if (mg.el.wsize == 0) {
    dvec<uint32_t> dout;
    global_degree(h, c, mg, mg.el.s.data(), dout);
    hipLaunchKernelGGL(k_u32_to_f64, grid_for(mg.vrange, kBlock, 4096), kBlock, 0, h.stream, (uint32_t const*)dout.data(), mg.vrange, outw.data());
    h.sync();
  } else {
    HIP_TRY(hipMemsetAsync(outw.data(), 0, (size_t)mg.vrange * 8, h.stream));
    if (mg.el.n > 0) {
      if (mg.el.wsize == 4) hipLaunchKernelGGL(k_add_weights_f64<float>, grid_for(mg.el.n, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)mg.el.s.data(), mg.el.w.as<float const>(), mg.el.n, mg.vmin, outw.data());
      else hipLaunchKernelGGL(k_add_weights_f64<double>, grid_for(mg.el.n, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)mg.el.s.data(), mg.el.w.as<double const>(), mg.el.n, mg.vmin, outw.data());
    }
    c.all_reduce_sum_f64(h, outw.data(), mg.vrange);
  }
  vertex_order_t vo;
  degree_order(h, din.data(), mg.present.data(), mg.vrange, mg.nv_global, vo);
  din = dvec<uint32_t>();
  tr.step("degree order");
  int64_t const nvg = mg.nv_global;
  int64_t const L   = std::max<int64_t>((nvg + P - 1) / P, 1);
  CGA_EXPECTS((int64_t)std::max(R, C) * L < ((int64_t)1 << 31), CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "2-D multi-GPU PageRank: a block's rows / columns must fit 31 bits");
  part->L     = L;
  part->n_own = nvg > me ? (nvg - me + P - 1) / P : 0;
  size_t const n1 = (size_t)std::max<int64_t>(part->n_own, 1);
  part->local_vertices.resize_discard(n1);
  bool const f64 = mg.el.wsize == 8;
  part->outw_own.alloc((size_t)L * (f64 ? 8 : 4));
  HIP_TRY(hipMemsetAsync(part->outw_own.ptr, 0, (size_t)L * (f64 ? 8 : 4), h.stream));
  if (part->n_own > 0) {
    hipLaunchKernelGGL(k_owned_ids, grid_for(part->n_own, kBlock, 4096), kBlock, 0, h.stream, (uint32_t const*)vo.order.data(), part->n_own, P, me, mg.vmin, part->local_vertices.data());
    if (f64) hipLaunchKernelGGL((k_take_owned<double, double>), grid_for(part->n_own, kBlock, 4096), kBlock, 0, h.stream, (double const*)outw.data(), (uint32_t const*)vo.order.data(), part->n_own, P, me, part->outw_own.as<double>());
    else hipLaunchKernelGGL((k_take_owned<float, double>), grid_for(part->n_own, kBlock, 4096), kBlock, 0, h.stream, (double const*)outw.data(), (uint32_t const*)vo.order.data(), part->n_own, P, me, part->outw_own.as<float>());
  }
  h.sync();
  outw = dvec<double>();
  // every edge to the rank of its block
  int64_t const m = mg.el.n;
  size_t const m1 = (size_t)std::max<int64_t>(m, 1);
  std::vector<dev_buf> got;
  int64_t e_loc = 0;
  {
    dvec<int32_t> owner(m1), a(m1), b(m1);
    if (m > 0) hipLaunchKernelGGL(k_route2d, grid_for(m, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)mg.el.s.data(), (int32_t const*)mg.el.d.data(), m, mg.vmin, (int32_t const*)vo.pos.data(), P, R, L,
                                  owner.data(), a.data(), b.data());
    std::vector<column_t> cols{{a.data(), 4}, {b.data(), 4}};
    if (mg.el.wsize) cols.push_back({mg.el.w.ptr, mg.el.wsize});
    e_loc = shuffle_by_owner(h, c, owner.data(), m, cols, got);
  }
  tr.step("edge shuffle");
  CGA_EXPECTS(e_loc <= kMaxSignedEdges, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "multi-GPU PageRank: a rank's share must hold fewer than 2^31 edges");
  part->ne_local = e_loc;
  {
    int64_t const nverts = (int64_t)std::max(R, C) * L;
    dvec<int32_t> verts((size_t)nverts);
    iota_i32(h, verts.data(), nverts, 0);
    h.sync();
    device_array_view_t vv{verts.data(), (size_t)nverts, INT32}, sv{got[0].ptr, (size_t)e_loc, INT32}, dv{got[1].ptr, (size_t)e_loc, INT32};
    device_array_view_t wv{mg.el.wsize ? got[2].ptr : nullptr, (size_t)e_loc, f64 ? FLOAT64 : FLOAT32};
    cugraph_graph_properties_t props{FALSE, TRUE};
    cugraph_error_t* err = nullptr;
    cugraph_resource_handle_t const* hh = reinterpret_cast<cugraph_resource_handle_t const*>(&h);
    cugraph_error_code_t const rc = cugraph_graph_create_sg(hh, &props, reinterpret_cast<cugraph_type_erased_device_array_view_t const*>(&vv),
                                                            reinterpret_cast<cugraph_type_erased_device_array_view_t const*>(&sv),
                                                            reinterpret_cast<cugraph_type_erased_device_array_view_t const*>(&dv),
                                                            mg.el.wsize ? reinterpret_cast<cugraph_type_erased_device_array_view_t const*>(&wv) : nullptr, nullptr, nullptr, TRUE, FALSE, FALSE,
                                                            FALSE, FALSE, FALSE, &part->local, &err);
    if (rc != CUGRAPH_SUCCESS) {
      std::string msg = err ? cugraph_error_message(err) : "?";
      cugraph_error_free(err);
      throw api_error(rc, "2-D multi-GPU PageRank: building the local block failed: " + msg);
    }
  }
  // Hypersparse rows: the block's rows are C vertex partitions that see the sources of ONE column group each, so a row of global in-degree d
  // holds ~d / C edges here and most tail rows hold none.  The reference stores such partitions as CSR + DCSR when the hypersparse segment
  // exists, i.e. when minor_comm_size * hypersparse_threshold_ratio (0.5) exceeds 1 (graph_view.hpp:245-247, renumber_edgelist_impl.cuh:746-757);
  // here: from C >= 4 (CUGRAPH_AMD_MG_DCSR=1 / 0 forces / forbids it, every rank the same).  The block's rows are not in one degree order (C
  // partitions, each degree-descending), so all rows are listed (first = 0); PageRank's re-blocking walks the form directly.
  {
    char const* env = getenv("CUGRAPH_AMD_MG_DCSR");
    bool const dcs  = env ? atoi(env) != 0 : C >= 4;
    if (dcs) {
      graph_t& block = *reinterpret_cast<graph_t*>(part->local);
      if (block.csc.built) compress_hypersparse(h, block.csc, block.nv, 0);
      if (block.csr.built) compress_hypersparse(h, block.csr, block.nv, 0);
      tr.step("hypersparse rows");
    }
  }
  tr.step("local block");
  mg.pr2d = std::move(part);
  return *mg.pr2d;
}

// ------------------------------------------------------------------------------------------------ traversal partition
namespace {
// sorts the received (row, minor[, extra key]) tuples into a CSR: rows ascending, inside a row ascending `sort_minor`; returns offsets and
// the `take` column in that order (+ weights)
void build_local_csr(handle_t const& h, int32_t const* row, int32_t const* sort_minor, int32_t const* take, float const* w, int64_t m, int64_t n_rows,
                     dvec<int32_t>& offsets, dvec<int32_t>& indices, dvec<float>* weights, double const* w64 = nullptr, dvec<double>* weights64 = nullptr)
{
  size_t const m1 = (size_t)std::max<int64_t>(m, 1);
  offsets.resize_discard((size_t)n_rows + 2);
  indices.resize_discard(m1 + (size_t)kEdgePad);
  HIP_TRY(hipMemsetAsync(indices.data(), 0, (m1 + (size_t)kEdgePad) * 4, h.stream));
  if (weights) { weights->resize_discard(m1 + (size_t)kEdgePad); HIP_TRY(hipMemsetAsync(weights->data(), 0, (m1 + (size_t)kEdgePad) * 4, h.stream)); }
  if (weights64) { weights64->resize_discard(m1 + (size_t)kEdgePad); HIP_TRY(hipMemsetAsync(weights64->data(), 0, (m1 + (size_t)kEdgePad) * 8, h.stream)); }
  dvec<uint32_t> cnt((size_t)n_rows + 2);
  HIP_TRY(hipMemsetAsync(cnt.data(), 0, ((size_t)n_rows + 2) * 4, h.stream));
  if (m > 0) {
    histogram_i32(h, row, m, cnt.data(), n_rows);
    dvec<uint64_t> keys(m1), keys_tmp(m1);
    dvec<uint32_t> perm(m1), perm_tmp(m1);
    hipLaunchKernelGGL(k_pack_row_key, grid_for(m, kBlock, 8192), kBlock, 0, h.stream, row, sort_minor, m, keys.data(), perm.data());
    radix_sort_u64_u32(h, keys.data(), perm.data(), keys_tmp.data(), perm_tmp.data(), m, 0, 32 + bits_for((uint64_t)std::max<int64_t>(n_rows, 1)));
    gather_b32(h, reinterpret_cast<uint32_t const*>(take), perm.data(), reinterpret_cast<uint32_t*>(indices.data()), m);
    if (weights) gather_b32(h, reinterpret_cast<uint32_t const*>(w), perm.data(), reinterpret_cast<uint32_t*>(weights->data()), m);
    if (weights64) gather_b64(h, reinterpret_cast<uint64_t const*>(w64), perm.data(), reinterpret_cast<uint64_t*>(weights64->data()), m);
    h.sync();
  }
  exclusive_scan_u32(h, cnt.data(), reinterpret_cast<uint32_t*>(offsets.data()), n_rows + 1);
  h.sync();
}
}  // namespace

mg_traversal_part_t& mg_traversal_part(handle_t const& h, graph_t& g, bool weighted)
{
  mg_graph_t& mg = *g.mg;
  auto& slot     = mg.tr[weighted ? 1 : 0];
  if (slot) return *slot;
  comm_t& c   = *mg.comm;
  int const P = c.size, me = c.rank;
  CGA_EXPECTS(!weighted || mg.el.wsize != 0, CUGRAPH_INVALID_INPUT, "Graph must be weighted");
  bool const w64 = weighted && mg.el.wsize == 8;
  auto t       = std::make_unique<mg_traversal_part_t>();
  t->P         = P;
  t->rank      = me;
  t->nv_global = mg.nv_global;
  t->ne_global = mg.ne_global;
  t->has_weights = weighted;
  global_degree(h, c, mg, mg.el.s.data(), t->out_deg);
  vertex_order_t vo;
  degree_order(h, t->out_deg.data(), mg.present.data(), mg.vrange, mg.nv_global, vo);
  int64_t const nvg = mg.nv_global;
  t->n_rows = nvg > me ? (nvg - me + P - 1) / P : 0;
  t->L      = (((nvg + P - 1) / P) + 63) / 64 * 64;
  if (t->L == 0) t->L = 64;
  CGA_EXPECTS(t->L * P < ((int64_t)1 << 31), CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "multi-GPU traversal: compact global ids must fit 31 bits");
  t->local_vertices.resize_discard((size_t)std::max<int64_t>(t->n_rows, 1));
  if (t->n_rows > 0) hipLaunchKernelGGL(k_owned_ids, grid_for(t->n_rows, kBlock, 4096), kBlock, 0, h.stream, (uint32_t const*)vo.order.data(), t->n_rows, P, me, mg.vmin, t->local_vertices.data());
  int64_t const m = mg.el.n;
  size_t const m1 = (size_t)std::max<int64_t>(m, 1);
  std::vector<dev_buf> got;
  int64_t e_loc = 0;
  {
    dvec<int32_t> owner(m1), a(m1), b(m1);
    if (m > 0) hipLaunchKernelGGL(k_route, grid_for(m, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)mg.el.s.data(), (int32_t const*)mg.el.d.data(), m, mg.vmin, (int32_t const*)vo.pos.data(), P, t->L, 1, owner.data(), a.data(), b.data(), (int32_t*)nullptr);
    std::vector<column_t> cols{{a.data(), 4}, {b.data(), 4}};
    if (weighted) cols.push_back({mg.el.w.ptr, mg.el.wsize});
    e_loc = shuffle_by_owner(h, c, owner.data(), m, cols, got);
  }
  CGA_EXPECTS(e_loc <= kMaxSignedEdges, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "multi-GPU traversal: a rank's share must hold fewer than 2^31 edges");
  t->ne_local = e_loc;
  build_local_csr(h, got[0].as<int32_t const>(), got[1].as<int32_t const>(), got[1].as<int32_t const>(), weighted && !w64 ? got[2].as<float const>() : nullptr, e_loc, t->n_rows,
                  t->offsets, t->indices, weighted && !w64 ? &t->weights : nullptr, w64 ? got[2].as<double const>() : nullptr, w64 ? &t->weights64 : nullptr);
  t->pos = std::move(vo.pos);
  // (order is needed once more by the in-edge copy: keep the external ids of all compact global ids instead -- built there)
  slot = std::move(t);
  return *slot;
}

void mg_traversal_in_edges(handle_t const& h, graph_t& g, mg_traversal_part_t& t)
{
  if (t.has_in) return;
  mg_graph_t& mg = *g.mg;
  comm_t& c      = *mg.comm;
  int const P    = c.size;
  int64_t const m = mg.el.n;
  size_t const m1 = (size_t)std::max<int64_t>(m, 1);
  std::vector<dev_buf> got;
  int64_t e_in = 0;
  {
    dvec<int32_t> owner(m1), a(m1), b(m1), cc(m1);
    if (m > 0) hipLaunchKernelGGL(k_route, grid_for(m, kBlock, 8192), kBlock, 0, h.stream, (int32_t const*)mg.el.s.data(), (int32_t const*)mg.el.d.data(), m, mg.vmin, (int32_t const*)t.pos.data(), P, t.L, 2, owner.data(), a.data(), b.data(), cc.data());
    std::vector<column_t> cols{{a.data(), 4}, {b.data(), 4}, {cc.data(), 4}};
    e_in = shuffle_by_owner(h, c, owner.data(), m, cols, got);
  }
  CGA_EXPECTS(e_in <= kMaxSignedEdges, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION, "multi-GPU traversal: a rank's in-edge share must hold fewer than 2^31 edges");
  // neighbours in ascending order of their external id (the first frontier member of a row is then the minimum-external-id parent)
  build_local_csr(h, got[0].as<int32_t const>(), got[2].as<int32_t const>(), got[1].as<int32_t const>(), nullptr, e_in, t.n_rows, t.in_offsets, t.in_indices, nullptr);
  // external id of every compact global id: all-gather of the ranks' L-entry tables (-1 = padding)
  dvec<int32_t> mine((size_t)t.L);
  fill_i32(h, mine.data(), t.L, -1);
  if (t.n_rows > 0) HIP_TRY(hipMemcpyAsync(mine.data(), t.local_vertices.data(), (size_t)t.n_rows * 4, hipMemcpyDeviceToDevice, h.stream));
  t.ext_of_g.resize_discard((size_t)t.L * P);
  c.all_gather(h, mine.data(), (size_t)t.L * 4, t.ext_of_g.data());
  t.has_in = true;
}

}  // namespace cga
