// Internal object model behind the opaque C-ABI structs of include/cugraph_c/*.h.
// MI355X-native: plain HIP runtime objects (one stream per handle, hipMalloc'ed buffers), no RAFT / RMM /
// Thrust.  Counterparts in the reference: cpp/src/c_api/{resource_handle,error,array,graph}.hpp.
#pragma once

#include <hip/hip_runtime.h>

#include <cugraph_amd/extensions.h>
#include <cugraph_c/array.h>
#include <cugraph_c/centrality_algorithms.h>
#include <cugraph_c/error.h>
#include <cugraph_c/graph.h>
#include <cugraph_c/graph_functions.h>
#include <cugraph_c/resource_handle.h>
#include <cugraph_c/traversal_algorithms.h>

#include <cstdint>
#include <cstdio>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace cga {

// ------------------------------------------------------------------------------------------ errors
struct api_error : std::runtime_error {
  cugraph_error_code_t code;
  api_error(cugraph_error_code_t c, std::string const& m) : std::runtime_error(m), code(c) {}
};

struct err_obj_t {  // behind cugraph_error_t (cpp/src/c_api/error.hpp)
  std::string message;
};

#define CGA_EXPECTS(cond, code, msg)                    \
  do {                                                  \
    if (!(cond)) throw ::cga::api_error((code), (msg)); \
  } while (0)

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      (void)hipGetLastError();                                                                     \
      throw ::cga::api_error(e_ == hipErrorOutOfMemory ? CUGRAPH_ALLOC_ERROR : CUGRAPH_UNKNOWN_ERROR, \
                             std::string("HIP error: ") + hipGetErrorString(e_) + " at " + __FILE__ + \
                               ":" + std::to_string(__LINE__));                                    \
    }                                                                                              \
  } while (0)

// Runs `f`, converts any exception into (code, *error) -- the reference's run_algorithm try/catch
// (cpp/src/c_api/utils.hpp).
template <typename F>
cugraph_error_code_t guarded(cugraph_error_t** error, F&& f)
{
  if (error) *error = nullptr;
  try {
    f();
    return CUGRAPH_SUCCESS;
  } catch (api_error const& e) {
    if (error) *error = reinterpret_cast<cugraph_error_t*>(new err_obj_t{e.what()});
    return e.code;
  } catch (std::bad_alloc const&) {
    if (error) *error = reinterpret_cast<cugraph_error_t*>(new err_obj_t{"host allocation failed"});
    return CUGRAPH_ALLOC_ERROR;
  } catch (std::exception const& e) {
    if (error) *error = reinterpret_cast<cugraph_error_t*>(new err_obj_t{e.what()});
    return CUGRAPH_UNKNOWN_ERROR;
  }
}

// ------------------------------------------------------------------------------------------ dtypes
inline size_t dtype_size(cugraph_data_type_id_t t)
{
  switch (t) {
    case INT8:
    case UINT8:
    case BOOL: return 1;
    case INT16:
    case UINT16: return 2;
    case INT32:
    case UINT32:
    case FLOAT32: return 4;
    case INT64:
    case UINT64:
    case FLOAT64:
    case SIZE_T: return 8;
    default: return 0;
  }
}

// ---------------------------------------------------------------------------------- device memory
// Owning device buffer on top of a process-wide caching pool (core.hip: pool_alloc / pool_free).  hipMalloc / hipFree are
// synchronous and slow for the sizes of this library (a BFS call used to spend more time in its dozen hipMalloc / hipFree
// pairs than in two of its levels; a PageRank plan build waited seconds for the driver after 60 GB of temporaries had just
// been freed), so freed blocks are kept and handed out again: best fit within 25 % slack, at most CUGRAPH_AMD_POOL_MAX_GB
// (default 128, at most 45 % of the device) cached, everything is released and the request retried when hipMalloc fails;
// cugraph_amd_memory_pool_trim / _trim_large hand cached blocks back for other allocators of the process (torch, cupy, RMM).
// Stream order: every API entry point names its handle's stream (H() -> pool_set_stream);
// a freed block records an event on that stream, and a reuse from a different stream waits for it (a reuse on the same stream
// is ordered by the stream).  CUGRAPH_AMD_POOL=0 turns the pool off; CUGRAPH_AMD_POOL_DEBUG=1 reports blocks that are freed
// while their stream still has work queued.
void* pool_alloc(size_t n_bytes, size_t* granted);
void pool_free(void* ptr, size_t granted) noexcept;
void pool_set_stream(hipStream_t s) noexcept;
void pool_forget_stream(hipStream_t s) noexcept;
size_t pool_release_large_blocks(size_t block_bytes = (size_t)256 << 20) noexcept;

struct dev_buf {
  void* ptr{nullptr};
  size_t bytes{0};
  size_t granted{0};  // what the pool handed out (>= bytes)
  dev_buf() = default;
  explicit dev_buf(size_t n_bytes) { alloc(n_bytes); }
  dev_buf(dev_buf const&)            = delete;
  dev_buf& operator=(dev_buf const&) = delete;
  dev_buf(dev_buf&& o) noexcept : ptr(o.ptr), bytes(o.bytes), granted(o.granted) { o.ptr = nullptr; o.bytes = 0; o.granted = 0; }
  dev_buf& operator=(dev_buf&& o) noexcept
  {
    if (this != &o) { release(); ptr = o.ptr; bytes = o.bytes; granted = o.granted; o.ptr = nullptr; o.bytes = 0; o.granted = 0; }
    return *this;
  }
  ~dev_buf() { release(); }
  void alloc(size_t n_bytes)
  {
    release();
    if (n_bytes == 0) return;
    ptr   = pool_alloc(n_bytes, &granted);  // throws api_error(CUGRAPH_ALLOC_ERROR)
    bytes = n_bytes;
  }
  void release()
  {
    if (ptr) pool_free(ptr, granted);
    ptr = nullptr; bytes = 0; granted = 0;
  }
  template <typename T> T* as() const { return static_cast<T*>(ptr); }
};

template <typename T>
struct dvec {  // typed convenience wrapper
  dev_buf buf;
  size_t n{0};
  dvec() = default;
  explicit dvec(size_t n_) : buf(n_ * sizeof(T)), n(n_) {}
  T* data() const { return buf.as<T>(); }
  size_t size() const { return n; }
  void resize_discard(size_t n_) { buf.alloc(n_ * sizeof(T)); n = n_; }
};

// ------------------------------------------------------------------------------------------ arrays
struct device_array_view_t {  // behind cugraph_type_erased_device_array_view_t (c_api/array.hpp)
  void* data;
  size_t size;
  cugraph_data_type_id_t type;
  template <typename T> T* as() const { return static_cast<T*>(data); }
};

struct device_array_t {  // behind cugraph_type_erased_device_array_t
  dev_buf buf;
  size_t size{0};
  cugraph_data_type_id_t type{INT32};
  device_array_t(size_t n, cugraph_data_type_id_t t) : buf(n * dtype_size(t)), size(n), type(t) {}
  device_array_t(dev_buf&& b, size_t n, cugraph_data_type_id_t t) : buf(std::move(b)), size(n), type(t) {}
  device_array_view_t* new_view() { return new device_array_view_t{buf.ptr, size, type}; }
};

struct host_array_view_t {
  void* data;
  size_t size;
  cugraph_data_type_id_t type;
};
struct host_array_t {
  struct free_deleter { void operator()(uint8_t* q) const { std::free(q); } };
  using storage_t = std::unique_ptr<uint8_t[], free_deleter>;
  storage_t data;
  size_t size;
  cugraph_data_type_id_t type;
};

// ------------------------------------------------------------------------------------------ handle
struct kernel_timer {  // HIP-event pairs recorded on the handle's stream (extensions.h)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
};

struct handle_t {  // behind cugraph_resource_handle_t
  int device{0};
  hipStream_t stream{nullptr};
  hipStream_t own_stream{nullptr};  // the stream created with the handle (destroyed with it); `stream` may be a borrowed one
  bool stream_borrowed{false};      // the caller's stream is shared: stream order replaces the host syncs of the stepping APIs
  int num_cus{256};
  size_t lds_per_block{65536};
  int rank{0};
  int comm_size{1};
  void* comm{nullptr};    // cga::comm_t (comm.hpp) when the handle was created on a communicator; borrowed, not owned
  void* pinned{nullptr};  // 64 KiB pinned scratch for scalar read-backs
  bool timing{false};
  std::map<std::string, kernel_timer> timers;
  int pagerank_hot_tile{-1};  // -1 = auto
  cugraph_amd_traversal_stats_t last_stats{};
  // A second stream for work that has no place on an algorithm's critical path (e.g. the vertex column of a traversal result, copied while the
  // traversal runs): side_fork() makes it wait for everything enqueued on `stream` so far and returns it, side_join() makes `stream` wait for it.
  // Created on first use, destroyed with the handle.
  mutable hipStream_t side_stream{nullptr};
  mutable hipEvent_t side_ev[2]{nullptr, nullptr};
  hipStream_t side_fork() const
  {
    if (!side_stream) {
      HIP_TRY(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&side_ev[0], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&side_ev[1], hipEventDisableTiming));
    }
    HIP_TRY(hipEventRecord(side_ev[0], stream));
    HIP_TRY(hipStreamWaitEvent(side_stream, side_ev[0], 0));
    return side_stream;
  }
  void side_join() const
  {
    if (!side_stream) return;
    HIP_TRY(hipEventRecord(side_ev[1], side_stream));
    HIP_TRY(hipStreamWaitEvent(stream, side_ev[1], 0));
  }

  template <typename T> T* pinned_as() const { return static_cast<T*>(pinned); }
  void sync() const { HIP_TRY(hipStreamSynchronize(stream)); }
  // D2H of a few scalars through the pinned page; synchronises.
  template <typename T> void read_back(T* host_out, T const* dev_src, size_t n) const
  {
    HIP_TRY(hipMemcpyAsync(pinned, dev_src, n * sizeof(T), hipMemcpyDeviceToHost, stream));
    sync();
    std::memcpy(host_out, pinned, n * sizeof(T));
  }
};

// CUGRAPH_AMD_BUILD_TRACE=1: wall time of the steps of graph / plan construction on stderr (each step ends with a stream
// sync, so host-side costs -- allocations, read-backs -- are inside the step that incurs them)
struct build_trace {
  handle_t const& h;
  char const* scope;
  bool on;
  std::chrono::steady_clock::time_point t0, t;
  build_trace(handle_t const& h_, char const* scope_) : h(h_), scope(scope_), on(getenv("CUGRAPH_AMD_BUILD_TRACE") != nullptr)
  {
    if (on) { h.sync(); t0 = t = std::chrono::steady_clock::now(); }
  }
  void step(char const* what)
  {
    if (!on) return;
    h.sync();
    auto const now = std::chrono::steady_clock::now();
    fprintf(stderr, "[build] %-12s %-28s %8.2f ms\n", scope, what, std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
  ~build_trace()
  {
    if (on) fprintf(stderr, "[build] %-12s %-28s %8.2f ms\n", scope, "TOTAL", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
};

struct timed_launch {  // RAII bracket: records start/stop events when timing is enabled
  handle_t* h;
  kernel_timer* t{nullptr};
  hipEvent_t stop{nullptr};
  hipStream_t s{nullptr};  // the stream the bracketed launch goes to (default: the handle's)
  timed_launch(handle_t const& hc, char const* family, hipStream_t stream = nullptr) : h(const_cast<handle_t*>(&hc)), s(stream ? stream : hc.stream)
  {
    if (!h->timing) return;
    t = &h->timers[family];
    hipEvent_t start;
    (void)hipEventCreate(&start);
    (void)hipEventCreate(&stop);
    (void)hipEventRecord(start, s);
    t->events.emplace_back(start, stop);
  }
  ~timed_launch()
  {
    if (t) (void)hipEventRecord(stop, s);
  }
};

struct tiled_csc_t;  // spmv_tiled.hpp
struct mg_graph_t;   // mg_graph.hpp

// ------------------------------------------------------------------------------------------- graph
// One compressed-sparse orientation.  `major` = row vertex: source for CSR (store_transposed = false),
// destination for CSC (store_transposed = true).  Neighbour lists ascending, multi-edges kept.
// Rows are processed through a degree-descending schedule: row_order == nullptr means rows are already
// numbered by descending degree (renumber = TRUE, primary orientation).
// CSR + DCSR hybrid row storage (DCSC for a transposed orientation): the reference's compress_hypersparse_offsets
// (cpp/src/structure/detail/structure_utils.cuh:139-195) and the dcs_nzd_vertices / major_hypersparse_first half of
// edge_partition_device_view_t (cpp/include/cugraph/edge_partition_device_view.cuh:43-58, 820-835).  Rows [0, first) keep one offset each; a row
// >= first is stored only when it has an edge: nzd[] lists those rows in ascending order and offsets[] has first + n_nzd + 1 entries.
// Where it pays here: orientations whose ids are NOT degree-sorted and whose rows are mostly empty -- the local block of the 2-D multi-GPU
// PageRank layout (rows of C vertex partitions that see 1 / C of the sources each; mg_graph.hip) and renumber = FALSE graphs over a sparse id
// range.  An orientation in this form has NO plain offsets array; consumers walk it through rows_view_t (below) or ask ensure_orientation for
// the plain form (which re-inflates it).
struct hypersparse_t {
  int64_t first{-1};      // -1: the orientation is in plain form
  int64_t n_nzd{0};
  dvec<int32_t> nzd;      // [n_nzd] rows >= first with at least one edge, ascending
  dvec<int32_t> offsets;  // [first + n_nzd + 1] edge positions (unsigned 32-bit words, as orientation_t::offsets)
  bool active() const { return first >= 0; }
  int64_t n_stored() const { return first + n_nzd; }
};

// What a kernel needs to walk the STORED rows of an orientation in either form: stored index k in [0, n_stored) <-> row row_of(k), edges
// [offsets[k], offsets[k + 1]).  Plain form: nzd == nullptr, k == row.
struct rows_view_t {
  int32_t const* offsets{nullptr};
  int32_t const* nzd{nullptr};
  int64_t first{0};
  int64_t n_stored{0};
  __host__ __device__ bool plain() const { return nzd == nullptr; }
  __device__ int32_t row_of(int64_t k) const { return (nzd == nullptr || k < first) ? (int32_t)k : nzd[k - first]; }
  // stored index of the first stored row >= r (n_stored when there is none)
  __device__ int64_t lower_bound(int64_t r) const
  {
    if (nzd == nullptr || r <= first) return r < n_stored ? r : n_stored;
    int64_t lo = 0, hi = n_stored - first;
    while (lo < hi) {
      int64_t const mid = (lo + hi) >> 1;
      if ((int64_t)nzd[mid] < r) lo = mid + 1; else hi = mid;
    }
    return first + lo;
  }
  // stored index of row r, or -1 when the row is not stored (it has no edge): major_hypersparse_idx_from_major_nocheck
  __device__ int64_t find(int64_t r) const
  {
    if (nzd == nullptr || r < first) return r < n_stored ? r : -1;
    int64_t const k = lower_bound(r);
    return (k < n_stored && (int64_t)nzd[k - first] == r) ? k : -1;
  }
  __device__ uint32_t degree_of_row(int64_t r) const
  {
    int64_t const k = find(r);
    return k < 0 ? 0u : (uint32_t)offsets[k + 1] - (uint32_t)offsets[k];
  }
};

struct orientation_t {
  bool built{false};
  dvec<int32_t> offsets;    // V + 1 (empty while the orientation is in hypersparse form: dcs)
  hypersparse_t dcs;        // CSR + DCSR hybrid form of the rows, see hypersparse_t
  dvec<int32_t> indices;    // E (minor ids)
  dev_buf weights;          // E * sizeof(weight) or empty
  dev_buf edge_ids;         // E * (4 or 8) bytes or empty: the caller's edge ids in this orientation's edge order (graph_t::edge_id_type)
  dvec<int32_t> edge_types; // E or empty: the caller's edge type ids in this orientation's edge order
  dvec<int32_t> row_order;  // V (permutation, degree descending) or empty = identity
  // seg[k] = number of scheduled rows with degree >= seg_threshold[k]; seg[4] = number of non-empty rows
  static constexpr int n_seg = 5;
  int64_t seg[n_seg]{0, 0, 0, 0, 0};
  int32_t max_degree{0};
  // bit e set <=> edge position e is the first edge of a row (built lazily; used by the edge-balanced
  // PageRank kernel, which needs the non-empty rows to be the id prefix [0, seg[4]))
  dvec<uint32_t> rowstart_bits;
  // column-tiled re-blocking of this orientation (built lazily by the PageRank plan, cached for later calls)
  std::shared_ptr<tiled_csc_t> tiled;
  // row (major id) of every edge position: built lazily by the first SSSP whose frontier holds a large share of the graph's edges
  // (k_sssp_sweep streams the whole edge list then: 4 more bytes per edge position, no row gathers); E + pad or empty
  dvec<int32_t> edge_rows;
};

constexpr int32_t kSegThreshold[orientation_t::n_seg] = {4096, 64, 16, 4, 1};
constexpr int64_t kEdgePad = 2048;  // indices / weights are over-allocated so 16-byte tail loads stay in bounds
// Edge positions are unsigned 32-bit words (the offsets arrays are declared int32_t and read as uint32_t where a graph may have
// 2^31 or more edges: graph construction, degrees, BFS / SSSP, and -- round 5 -- the column-tiled PageRank plan).  PageRank's single-pass
// comparison kernels (signed offsets) and Louvain keep the signed limit: kMaxSignedEdges.
constexpr int64_t kMaxGraphEdges  = ((int64_t)1 << 32) - 4097;
constexpr int64_t kMaxSignedEdges = ((int64_t)1 << 31) - 1;

// INT64 / sparse external ids are translated at the C-API boundary (outer_ids.hip): compact id c <-> ext[c], ext ascending
struct outer_ids_t {
  bool active{false};
  bool identity{false};                     // ext[c] == c (renumber = FALSE graphs: only the id TYPE differs)
  cugraph_data_type_id_t type{INT32};       // the vertex type the caller sees
  dvec<int64_t> ext;
};

struct graph_t {  // behind cugraph_graph_t (cpp/src/c_api/graph.hpp:61-77)
  outer_ids_t outer;
  cugraph_data_type_id_t api_vertex_type() const { return outer.active ? outer.type : vertex_type; }
  cugraph_data_type_id_t vertex_type{INT32};
  cugraph_data_type_id_t edge_type{INT32};
  cugraph_data_type_id_t weight_type{FLOAT32};
  bool store_transposed{false};  // orientation requested at creation (primary)
  bool has_weights{false};
  bool has_edge_ids{false}, has_edge_types{false};  // edge properties carried for cugraph_decompress_to_edgelist (graph_sg.cpp:781-830)
  cugraph_data_type_id_t edge_id_type{INT32};
  bool renumbered{false};
  cugraph_graph_properties_t props{FALSE, FALSE};
  int64_t nv{0};
  int64_t ne{0};
  dvec<int32_t> number_map;  // internal -> external, size V
  // external -> internal: dense table over [ext_min, ext_min + ext_range) (-1 = absent); empty when the
  // graph is not renumbered (identity).
  int64_t ext_min{0};
  dvec<int32_t> ext2int;
  orientation_t csr;  // by source
  orientation_t csc;  // by destination
  // cached out-weight sums (a5 in SURVEY 8a): float or double, size V
  dev_buf out_weight_sums;
  bool out_weight_sums_valid{false};
  double weight_sum{0};  // sum of the edge weights (SSSP bucket width), cached
  bool weight_sum_valid{false};
  bool weights_uniform{false};  // every edge weight is the same value (found with the sum): distance order inside a round carries no information
  int64_t sssp_heavy_cut{-1};  // SSSP (radix sub-queues): ids below this have at least average out-degree (ids are degree-sorted); -1 = not counted yet
  int bfs_calls{0};  // the CSC (bottom-up BFS levels) is built from the second traversal of a non-symmetric graph on
  // cugraph_graph_create_mg on a communicator handle: this rank's slice + the partitions built from it (mg_graph.hpp); the single-GPU
  // members above (orientations, number_map) stay empty, nv / ne are the GLOBAL counts
  std::shared_ptr<mg_graph_t> mg;
};

inline handle_t const& H(cugraph_resource_handle_t const* h)
{
  CGA_EXPECTS(h != nullptr, CUGRAPH_INVALID_HANDLE, "resource handle is NULL");
  handle_t const& hh = *reinterpret_cast<handle_t const*>(h);
  pool_set_stream(hh.stream);  // frees and reuses of device blocks made by this call are ordered on this stream
  return hh;
}
inline graph_t& GM(cugraph_graph_t* g)  // entry points that also take multi-GPU graphs
{
  CGA_EXPECTS(g != nullptr, CUGRAPH_INVALID_INPUT, "graph is NULL");
  return *reinterpret_cast<graph_t*>(g);
}
inline graph_t& G(cugraph_graph_t* g)
{
  graph_t& gg = GM(g);
  CGA_EXPECTS(!gg.mg, CUGRAPH_NOT_IMPLEMENTED, "this entry point does not take a multi-GPU graph in this build (PageRank, BFS, SSSP, Louvain, has_vertex do)");
  return gg;
}
inline device_array_view_t const* V(cugraph_type_erased_device_array_view_t const* v)
{
  return reinterpret_cast<device_array_view_t const*>(v);
}

// results
struct centrality_result_t {  // c_api/centrality_result.hpp
  device_array_t* vertex_ids;
  device_array_t* values;
  size_t num_iterations;
  bool converged;
};
struct coo_t {  // behind cugraph_coo_t (c_api/coo.hpp)
  device_array_t* src{nullptr};
  device_array_t* dst{nullptr};
  device_array_t* wgt{nullptr};
  device_array_t* ids{nullptr};    // cugraph_generate_edge_ids
  device_array_t* types{nullptr};  // cugraph_generate_edge_types
  ~coo_t() { delete src; delete dst; delete wgt; delete ids; delete types; }
};
struct coo_list_t {  // behind cugraph_coo_list_t: owns its elements
  std::vector<coo_t*> list;
  ~coo_list_t() { for (auto* c : list) delete c; }
};

struct paths_result_t {  // c_api/paths_result.hpp
  device_array_t* vertex_ids;
  device_array_t* distances;
  device_array_t* predecessors;
};

// ----------------------------------------------------------------- device primitives (prims.hip)
constexpr int kBlock = 256;
inline int grid_for(int64_t n, int block = kBlock, int64_t cap = 1 << 20)
{
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return static_cast<int>(g);
}

void fill_i32(handle_t const& h, int32_t* p, int64_t n, int32_t v);
void fill_u32(handle_t const& h, uint32_t* p, int64_t n, uint32_t v);
void fill_f32(handle_t const& h, float* p, int64_t n, float v);
void fill_f64(handle_t const& h, double* p, int64_t n, double v);
void iota_i32(handle_t const& h, int32_t* p, int64_t n, int32_t first);
// min / max of an int32 array (n > 0); result on host (synchronises)
void minmax_i32(handle_t const& h, int32_t const* p, int64_t n, int32_t* mn, int32_t* mx);
// exclusive prefix sum, out[i] = sum_{j<i} in[j]; in == out allowed; returns nothing (total at out[n] if
// the caller sized out as n + 1 and passes n + 1 with in[n] = 0).
void exclusive_scan_u32(handle_t const& h, uint32_t const* in, uint32_t* out, int64_t n);
// histogram: counts[keys[i]] += 1 (counts pre-zeroed by the caller)
// `range` (number of counters) lets large inputs take the partitioned path of prims.hip; 0 = unknown / small
void histogram_i32(handle_t const& h, int32_t const* keys, int64_t n, uint32_t* counts, int64_t range = 0);
// counts[rank[keys[i] - vmin]] += 1 (rank == nullptr: counts[keys[i]] += 1)
void histogram_i32_mapped(handle_t const& h, int32_t const* keys, int64_t n, int64_t vmin, uint32_t const* rank, uint32_t* counts,
                          int64_t range = 0);
// stable LSD radix sort of 64-bit keys (only bits [bit_lo, bit_hi) are examined) with a 32-bit payload.
// keys/vals are sorted in place; tmp buffers of the same size are required.
void radix_sort_u64_u32(handle_t const& h, uint64_t* keys, uint32_t* vals, uint64_t* keys_tmp,
                        uint32_t* vals_tmp, int64_t n, int bit_lo, int bit_hi);
// one stable 8-bit pass without synchronisation or copy-back: the result is in (keys_out, vals_out); hist = caller's scratch of radix_pass_scratch(n) words
size_t radix_pass_scratch(int64_t n);
void radix_pass_u64_u32(handle_t const& h, uint64_t const* keys_in, uint32_t const* vals_in, uint64_t* keys_out, uint32_t* vals_out, int64_t n, int shift, int bits,
                        uint32_t* hist);
// out[i] = src[idx[i]] for 4- / 8-byte elements
void gather_b32(handle_t const& h, uint32_t const* src, uint32_t const* idx, uint32_t* out, int64_t n);
void gather_b64(handle_t const& h, uint64_t const* src, uint32_t const* idx, uint64_t* out, int64_t n);

// edge-list preprocessing behind the graph-creation flags (edgelist.hip); ids are EXTERNAL ids in [vmin, vmin + vrange)
struct edge_list_t {
  dvec<int32_t> s, d;
  dev_buf w;         // optional weights (wsize bytes each)
  size_t wsize{0};
  int64_t n{0};
};
void edgelist_drop_self_loops(handle_t const& h, edge_list_t& el);
void edgelist_drop_multi_edges(handle_t const& h, edge_list_t& el, int64_t vmin, int64_t vrange);
void edgelist_symmetrize(handle_t const& h, edge_list_t& el, int64_t vmin, int64_t vrange);
// do_expensive_check helpers (create_graph_from_edgelist_impl.cuh:72-126, 261-333)
bool edgelist_is_symmetric(handle_t const& h, edge_list_t const& el, int64_t vmin, int64_t vrange);
bool edgelist_has_parallel_edges(handle_t const& h, edge_list_t const& el, int64_t vmin, int64_t vrange);
bool vertex_list_has_duplicates(handle_t const& h, int32_t const* v, int64_t n, int64_t vmin, int64_t vrange);

// graph construction (graph.hip)
// dcs_aware = false: the caller reads orientation_t::offsets directly, so an orientation in hypersparse form is re-inflated first
void ensure_orientation(handle_t const& h, graph_t& g, bool transposed, bool dcs_aware = false);
// hypersparse rows (graph.hip): rows >= first that have no edge lose their offset (offsets is released); no-op on an orientation already in that form
void compress_hypersparse(handle_t const& h, orientation_t& o, int64_t nv, int64_t first);
void inflate_offsets(handle_t const& h, orientation_t& o, int64_t nv);  // back to the plain form (dcs is released)
inline rows_view_t rows_view(orientation_t const& o, int64_t nv)
{
  rows_view_t v;
  if (o.dcs.active()) { v.offsets = o.dcs.offsets.data(); v.nzd = o.dcs.nzd.data(); v.first = o.dcs.first; v.n_stored = o.dcs.n_stored(); }
  else { v.offsets = o.offsets.data(); v.nzd = nullptr; v.first = nv; v.n_stored = nv; }
  return v;
}
// external -> internal ids (in place); absent ids become -1
void renumber_ext_to_int(handle_t const& h, graph_t const& g, int32_t* ids, int64_t n);
// internal -> external (in place); negative ids stay as they are
void unrenumber_int_to_ext(handle_t const& h, graph_t const& g, int32_t* ids, int64_t n);
int64_t count_negative_i32(handle_t const& h, int32_t const* ids, int64_t n);
int64_t count_negative_f32(handle_t const& h, float const* v, int64_t n);
int64_t count_negative_f64(handle_t const& h, double const* v, int64_t n);

// INT64 / sparse external ids at the C-API boundary (outer_ids.hip)
void outer_collect(handle_t const& h, device_array_view_t const* const* cols, int ncols, dvec<int64_t>& ext);
void outer_to_compact(handle_t const& h, outer_ids_t const& o, void const* ids, cugraph_data_type_id_t type, int64_t n, int32_t* out);
device_array_t* outer_from_compact(handle_t const& h, outer_ids_t const& o, int32_t const* ids, int64_t n);
void outer_replace_ids(handle_t const& h, graph_t const& g, device_array_t*& col);
void outer_replace_dist(handle_t const& h, graph_t const& g, device_array_t*& col);
void outer_narrow_dist(handle_t const& h, int64_t const* d, int64_t n, int32_t* out);
// A vertex-id column handed to an algorithm: checks its type against the graph's (CUGRAPH_INVALID_INPUT as bfs.cpp:198-205)
// and, for graphs with outer ids, replaces it by an owned column of compact int32 ids (-1 = not a vertex).
struct vertex_column_in {
  dvec<int32_t> owned;
  device_array_view_t view{nullptr, 0, INT32};
  device_array_view_t const* get(handle_t const& h, graph_t const& g, device_array_view_t const* v, char const* what)
  {
    if (v == nullptr) return nullptr;
    CGA_EXPECTS(v->type == g.api_vertex_type(), CUGRAPH_INVALID_INPUT, std::string("vertex type of graph and ") + what + " must match");
    if (!g.outer.active) return v;
    owned.resize_discard(v->size > 0 ? v->size : 1);
    outer_to_compact(h, g.outer, v->data, v->type, (int64_t)v->size, owned.data());
    h.sync();
    view = device_array_view_t{owned.data(), v->size, INT32};
    return &view;
  }
};

}  // namespace cga
