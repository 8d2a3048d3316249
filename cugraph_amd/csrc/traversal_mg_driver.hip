// cugraph_bfs / cugraph_sssp on a graph from cugraph_graph_create_mg (a handle on the library's communicator): the level loop of the
// partitioned traversals inside the library.  Per-rank compute = the plan of traversal_mg.hip (sender-side reduction of candidates,
// counting sort by owner, owner-side apply, bottom-up BFS levels on the in-edge copy); exchange = peer pushes into persistent windows
// (comm.hpp): the candidate tuples go straight into every owner's receive window at the offset the count matrix assigns (one host
// all-gather of P counts per level through the bootstrap segment), the L-bit new-frontier bitmaps and three statistics per rank into a
// [P][L/32] / [P][4] window double-buffered by level parity -- one signal per exchange, no collective launch.
// Replaces the multi_gpu = true halves of cpp/src/traversal/bfs_impl.cuh:133-870, sssp_impl.cuh:169-566 and the shuffle of
// prims/transform_reduce_if_v_frontier_outgoing_e_by_dst.cuh:981-1074 (what cugraph_amd/mg_traversal.py does over torch.distributed).
#include "comm.hpp"
#include "mg_graph.hpp"
#include "traversal_common.hpp"

#include <algorithm>
#include <cfloat>
#include <cstring>

namespace cga {

// traversal_mg.hip: the launch half and the host half of the plan's apply / bottom_up (the exported calls synchronise in between)
void mg_plan_expand_launch(cugraph_amd_traversal_mg_plan_t* plan);
unsigned long long const* mg_plan_totals(cugraph_amd_traversal_mg_plan_t* plan);
void mg_plan_apply_launch(cugraph_amd_traversal_mg_plan_t* plan, int32_t const* recv, size_t n_tuples, uint32_t level, uint32_t const* n_tuples_dev);
void mg_plan_bottom_up_launch(cugraph_amd_traversal_mg_plan_t* plan, uint32_t const* front, uint32_t level);
void const* mg_plan_counters(cugraph_amd_traversal_mg_plan_t* plan);
void mg_plan_adopt_level(cugraph_amd_traversal_mg_plan_t* plan, size_t n_next, unsigned long long out_edges, unsigned long long in_edges);

namespace {

void ck(cugraph_error_code_t rc, cugraph_error_t* err, char const* what)
{
  if (rc == CUGRAPH_SUCCESS) return;
  std::string msg = err ? cugraph_error_message(err) : "?";
  cugraph_error_free(err);
  throw api_error(rc, std::string(what) + ": " + msg);
}

__global__ void k_lookup_pos(int32_t const* ids, int64_t n, int64_t vmin, int64_t vrange, int32_t const* pos, uint32_t const* out_deg, int32_t* out_pos, unsigned long long* deg_sum)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t const k = (int64_t)ids[i] - vmin;
    int32_t const p = (k >= 0 && k < vrange) ? pos[k] : -1;
    out_pos[i]      = p;
    if (p >= 0 && deg_sum) atomicAdd(deg_sum, (unsigned long long)out_deg[k]);
  }
}

__global__ void k_sum_f32(float const* w, int64_t n, double* out)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  double s  = 0.0;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) s += (double)w[i];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0 && s != 0.0) atomicAdd(out, s);
}

__global__ void k_put_stats(unsigned long long* const* peer_s, int rank, int P, unsigned long long a, unsigned long long b, unsigned long long c)
{
  int const r = threadIdx.x;
  if (r < P) {
    unsigned long long* d = peer_s[r] + 4 * rank;
    d[0] = a; d[1] = b; d[2] = c; d[3] = 0;
  }
}
// the same from the level's device counters (one wavefront folds the replica lines: counters_t::fold without the trip to the host);
// the fourth word carries the length of the queue the level wrote (a bottom-up level's cross-check)
__global__ void k_put_stats_dev(unsigned long long* const* peer_s, int rank, int P, counters_t const* cnt)
{
  int const lane = threadIdx.x;
  unsigned long long found = 0, out = 0, in = 0;
  if (lane < CNT_REPLICAS) { found = cnt->rep[lane].n_found; out = cnt->rep[lane].out_edges; in = cnt->rep[lane].in_edges; }
  for (int o = 32; o; o >>= 1) {
    found += (unsigned long long)__shfl_xor((long long)found, o);
    out   += (unsigned long long)__shfl_xor((long long)out, o);
    in    += (unsigned long long)__shfl_xor((long long)in, o);
  }
  found += cnt->n_next; out += cnt->out_edges; in += cnt->in_edges;
  if (lane < P) {
    unsigned long long* d = peer_s[lane] + 4 * rank;
    d[0] = found; d[1] = out; d[2] = in; d[3] = cnt->n_big;
  }
}


// A top-down BFS level's all-to-all-v with no host in it (the reference: a device all-to-all of the counts, then of the tuples --
// cpp/include/cugraph/utilities/shuffle_comm.cuh:139-186): the bucket boundaries of `send` are read where the counting sort left them (totals, the list
// length in the counters), owner k's share goes to THIS rank's slot of k's window (slot = L tuples: a sender names a destination at most once per level,
// so no count matrix is needed to place it) and its length to word `rank` of k's count window.  blockIdx.y = owner.
__global__ void __launch_bounds__(256) k_push_tuples_dev(uint32_t const* send, unsigned long long const* totals, counters_t const* cnt, uint32_t* const* peer_slots,
                                                        uint32_t* const* peer_counts, int rank, int P, int64_t slot_words, int tw)
{
  int const k          = blockIdx.y;
  uint32_t const first = (uint32_t)totals[k];
  uint32_t const next  = k + 1 < P ? (uint32_t)(totals[k] >> 32) : cnt->n_next;
  int64_t const n      = (int64_t)(next - first) * tw;
  uint32_t const* src  = send + (int64_t)first * tw;
  uint32_t* dst        = peer_slots[k] + (int64_t)rank * slot_words;
  int64_t i            = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t const stride = (int64_t)gridDim.x * blockDim.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    uint32_t const a = src[i], b = src[i + stride], c = src[i + 2 * stride], e = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = e;
  }
  for (; i < n; i += stride) dst[i] = src[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) peer_counts[k][rank] = next - first;
}

}  // namespace

// the plan of one traversal family on one graph + its exchange windows (created on first use, freed with the graph: collective)
struct mg_traversal_run_t {
  comm_t* c{nullptr};
  handle_t const* h{nullptr};
  int mode{0}, tw{2};
  cugraph_amd_traversal_mg_plan_t* plan{nullptr};
  dvec<int32_t> send;
  size_t capacity{0};
  comm_window_t* twin{nullptr};                  // received candidate tuples, grouped by sender
  comm_window_t* bwin[2]{nullptr, nullptr};      // BFS: [P][L / 32] gathered new-frontier bits, by level parity
  comm_window_t* swin[2]{nullptr, nullptr};      // BFS: [P][4] (discoveries, their out- and in-degree sums)
  comm_window_t* cwin{nullptr};                  // BFS: [P] tuples every sender put into its slot of twin this level (written by the senders' devices)
  dvec<unsigned long long*> d_peer_s[2];
  dvec<uint32_t*> d_peer_t, d_peer_c;
  int channel{0};
  bool bottom_up_set{false};
  double delta{0.0};  // SSSP: width of the near / far window (32 x average weight / average degree over the whole graph; 0 = not computed yet)
  ~mg_traversal_run_t()
  {
    try {
      // (not h->stream: the resource handle of the first traversal may be gone by the time the graph is freed -- advisor finding, round 4)
      if (c) { (void)hipSetDevice(c->device); (void)hipDeviceSynchronize(); }
      if (plan) cugraph_amd_traversal_mg_plan_free(plan);
      if (cwin) c->window_free(cwin);
      for (int b = 1; b >= 0; --b) { if (swin[b]) c->window_free(swin[b]); if (bwin[b]) c->window_free(bwin[b]); }
      if (twin) c->window_free(twin);
      if (channel >= 2) c->channel_free(channel);
    } catch (...) {
    }
  }
};

namespace {

mg_traversal_run_t& ensure_run(handle_t const& h, graph_t& g, mg_traversal_part_t& t, int mode)
{
  if (t.run) {
    // the cached plan and its windows were built through the first traversal's resource handle; a later call may come with ANY handle of the same
    // communicator (the reference's contract: callers that create a handle per call): the run and its plan move to the handle of this call
    if (t.run->h != &h) {
      cugraph_amd_traversal_mg_plan_rebind(t.run->plan, reinterpret_cast<cugraph_resource_handle_t const*>(&h));
      t.run->h = &h;
    }
    return *t.run;
  }
  comm_t& c = *g.mg->comm;
  auto r    = std::make_shared<mg_traversal_run_t>();
  r->c = &c; r->h = &h; r->mode = mode; r->tw = mode == 0 ? 2 : 3;
  int const P = t.P;
  r->capacity = (size_t)std::max<int64_t>(std::min<int64_t>(t.L * P, std::max<int64_t>(t.ne_local, 1)), 1);
  r->send.resize_discard(r->capacity * r->tw + 64);
  cugraph_error_t* err = nullptr;
  auto hh = reinterpret_cast<cugraph_resource_handle_t const*>(&h);
  ck(cugraph_amd_traversal_mg_plan_create(hh, t.offsets.data(), t.indices.data(), t.has_weights ? t.weights.data() : nullptr, (size_t)t.n_rows, (size_t)t.ne_local, (size_t)t.L,
                                          t.rank, P, t.local_vertices.data(), mode, r->send.data(), r->capacity, &r->plan, &err),
     err, "multi-GPU traversal plan");
  cugraph_amd_traversal_mg_plan_keep_buffers(r->plan, TRUE);  // the gathered bitmaps live in persistent windows: no synchronisation per merge
  // every sender sends a destination at most once per level: at most L tuples per (sender, owner) pair
  r->channel = c.channel_alloc();
  r->twin    = c.window_create((size_t)P * (size_t)t.L * r->tw * 4);
  if (mode == 0) {
    for (int b = 0; b < 2; ++b) {
      r->bwin[b] = c.window_create((size_t)P * (size_t)(t.L / 32) * 4);
      r->swin[b] = c.window_create((size_t)P * 4 * sizeof(unsigned long long));
      HIP_TRY(hipMemsetAsync(r->bwin[b]->local, 0, (size_t)P * (size_t)(t.L / 32) * 4, h.stream));
      HIP_TRY(hipMemsetAsync(r->swin[b]->local, 0, (size_t)P * 4 * sizeof(unsigned long long), h.stream));
      r->d_peer_s[b].resize_discard(P);
      HIP_TRY(hipMemcpyAsync(r->d_peer_s[b].data(), r->swin[b]->peer.data(), (size_t)P * sizeof(void*), hipMemcpyHostToDevice, h.stream));
    }
    r->cwin = c.window_create((size_t)kCommMaxRanks * 4);
    HIP_TRY(hipMemsetAsync(r->cwin->local, 0, (size_t)kCommMaxRanks * 4, h.stream));
    r->d_peer_t.resize_discard(P);
    r->d_peer_c.resize_discard(P);
    HIP_TRY(hipMemcpyAsync(r->d_peer_t.data(), r->twin->peer.data(), (size_t)P * sizeof(void*), hipMemcpyHostToDevice, h.stream));
    HIP_TRY(hipMemcpyAsync(r->d_peer_c.data(), r->cwin->peer.data(), (size_t)P * sizeof(void*), hipMemcpyHostToDevice, h.stream));
  }
  h.sync();
  c.host_barrier();
  t.run = r;
  return *t.run;
}

// all-to-all-v of the candidate tuples `send` holds grouped by owner: returns the number of tuples now in the local receive window
size_t exchange_tuples(handle_t const& h, mg_traversal_run_t& r, int P, int me, size_t const* send_counts)
{
  comm_t& c = *r.c;
  std::vector<int64_t> mine(P), M((size_t)P * P);
  for (int k = 0; k < P; ++k) mine[k] = (int64_t)send_counts[k];
  c.host_allgather(mine.data(), (size_t)P * sizeof(int64_t), M.data());
  comm_push_desc_t d{};
  int64_t soff = 0, total = 0;
  for (int s = 0; s < P; ++s) total += M[(size_t)s * P + me];
  CGA_EXPECTS((size_t)total * r.tw * 4 <= r.twin->bytes[me], CUGRAPH_UNKNOWN_ERROR, "multi-GPU traversal: more candidates than the receive window holds");
  for (int k = 0; k < P; ++k) {
    int64_t roff = 0;  // where this rank's tuples start in owner k's window: behind the lower ranks'
    for (int s = 0; s < me; ++s) roff += M[(size_t)s * P + k];
    d.dst[k]   = r.twin->at<int32_t>(k) + roff * r.tw;
    d.src[k]   = r.send.data() + soff * r.tw;
    d.words[k] = mine[k] * r.tw;
    soff += mine[k];
  }
  d.n = P;
  c.push_multi(h.stream, d);
  c.wait(h.stream, r.channel, c.signal(h.stream, r.channel));
  return (size_t)total;
}

// the same for a BFS level without the host: counts and tuples travel from the device (k_push_tuples_dev); what the local window then holds is described by
// the count window -- the owner-side apply reads it there
void exchange_tuples_dev(handle_t const& h, mg_traversal_run_t& r, mg_traversal_part_t const& t)
{
  comm_t& c   = *r.c;
  int const P = t.P;
  // (at most L tuples per owner: enough workgroups for a full slot, few enough that an empty level costs one short launch)
  unsigned const gx = (unsigned)std::min<int64_t>(std::max<int64_t>((t.L * r.tw + 4095) / 4096, 1), std::max(h.num_cus * 8 / P, 8));
  hipLaunchKernelGGL(k_push_tuples_dev, dim3(gx, (unsigned)P), dim3(256), 0, h.stream, (uint32_t const*)r.send.data(), mg_plan_totals(r.plan),
                     static_cast<counters_t const*>(mg_plan_counters(r.plan)), (uint32_t* const*)r.d_peer_t.data(), (uint32_t* const*)r.d_peer_c.data(), t.rank, P,
                     (int64_t)t.L * r.tw, r.tw);
  c.wait(h.stream, r.channel, c.signal(h.stream, r.channel));
}

struct level_stats_t { unsigned long long n, out_sum, in_sum; };

// BFS: every rank's new-frontier bits + (discoveries, out-degree sum, in-degree sum) -> every rank; merges the bits into the visited set
level_stats_t share_frontier(handle_t const& h, mg_traversal_run_t& r, mg_traversal_part_t const& t, int b, level_stats_t mine, counters_t const* dev_counters = nullptr,
                             unsigned long long* own = nullptr /*[4]: this rank's (discoveries, out sum, in sum, queue length) as shipped*/)
{
  comm_t& c   = *r.c;
  int const P = t.P, me = t.rank;
  cugraph_error_t* err = nullptr;
  uint32_t const* bits = nullptr;
  ck(cugraph_amd_traversal_mg_plan_frontier_bits(r.plan, &bits, &err), err, "frontier bits");
  int64_t const W = t.L / 32;
  comm_push_desc_t d{};
  for (int k = 0; k < P; ++k) { d.dst[k] = r.bwin[b]->at<uint32_t>(k) + (int64_t)me * W; d.src[k] = bits; d.words[k] = W; }
  d.n = P;
  c.push_multi(h.stream, d);
  if (dev_counters) hipLaunchKernelGGL(k_put_stats_dev, 1, 64, 0, h.stream, (unsigned long long* const*)r.d_peer_s[b].data(), me, P, dev_counters);
  else hipLaunchKernelGGL(k_put_stats, 1, 64, 0, h.stream, (unsigned long long* const*)r.d_peer_s[b].data(), me, P, mine.n, mine.out_sum, mine.in_sum);
  c.wait(h.stream, r.channel, c.signal(h.stream, r.channel));
  ck(cugraph_amd_traversal_mg_plan_merge_visited(r.plan, static_cast<uint32_t const*>(r.bwin[b]->local), &err), err, "merge visited");  // (no synchronisation: keep_buffers)
  std::vector<unsigned long long> all((size_t)P * 4);
  HIP_TRY(hipMemcpyAsync(h.pinned, r.swin[b]->local, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, h.stream));
  h.sync();  // the level's one wait for the exchange: the statistics of all ranks are here, the bits are merged
  std::memcpy(all.data(), h.pinned, all.size() * sizeof(unsigned long long));
  c.check("multi-GPU BFS level");
  level_stats_t tot{0, 0, 0};
  for (int k = 0; k < P; ++k) { tot.n += all[4 * k]; tot.out_sum += all[4 * k + 1]; tot.in_sum += all[4 * k + 2]; }
  if (own) std::memcpy(own, all.data() + (size_t)4 * me, 4 * sizeof(unsigned long long));
  return tot;
}

// the union of the ranks' source lists (external ids), sorted, without duplicates: every rank ends with the same list
std::vector<int32_t> gather_sources(handle_t const& h, comm_t& c, device_array_view_t const* sources)
{
  int64_t const ns = sources ? (int64_t)sources->size : 0;
  std::vector<int32_t> mine((size_t)ns);
  if (ns > 0) HIP_TRY(hipMemcpyAsync(mine.data(), sources->data, (size_t)ns * 4, hipMemcpyDeviceToHost, h.stream));
  h.sync();
  std::vector<int64_t> counts(c.size);
  int64_t n64 = ns;
  c.host_allgather(&n64, sizeof(n64), counts.data());
  int64_t biggest = 0;
  for (auto x : counts) biggest = std::max(biggest, x);
  std::vector<int32_t> all;
  int64_t const chunk = (int64_t)(kCommSlotBytes / 4);
  std::vector<int32_t> buf((size_t)chunk), got((size_t)chunk * c.size);
  for (int64_t first = 0; first < biggest; first += chunk) {
    std::fill(buf.begin(), buf.end(), 0);
    for (int64_t k = 0; k < chunk && first + k < ns; ++k) buf[k] = mine[first + k];
    c.host_allgather(buf.data(), (size_t)chunk * 4, got.data());
    for (int r = 0; r < c.size; ++r)
      for (int64_t k = 0; k < chunk && first + k < counts[r]; ++k) all.push_back(got[(size_t)r * chunk + k]);
  }
  std::sort(all.begin(), all.end());
  all.erase(std::unique(all.begin(), all.end()), all.end());
  return all;
}

struct located_t {
  std::vector<int32_t> rows;  // local rows of the sources this rank owns
  unsigned long long out_deg_sum{0};
  int64_t n_sources{0};
};

located_t locate_sources(handle_t const& h, graph_t& g, mg_traversal_part_t const& t, std::vector<int32_t> const& ext, char const* api)
{
  mg_graph_t const& mg = *g.mg;
  located_t out;
  out.n_sources = (int64_t)ext.size();
  if (ext.empty()) return out;
  dvec<int32_t> d_ids(ext.size()), d_pos(ext.size());
  dvec<unsigned long long> d_sum(1);
  HIP_TRY(hipMemcpyAsync(d_ids.data(), ext.data(), ext.size() * 4, hipMemcpyHostToDevice, h.stream));
  HIP_TRY(hipMemsetAsync(d_sum.data(), 0, 8, h.stream));
  hipLaunchKernelGGL(k_lookup_pos, grid_for((int64_t)ext.size(), kBlock, 1024), kBlock, 0, h.stream, (int32_t const*)d_ids.data(), (int64_t)ext.size(), mg.vmin, mg.vrange,
                     (int32_t const*)t.pos.data(), (uint32_t const*)t.out_deg.data(), d_pos.data(), d_sum.data());
  std::vector<int32_t> pos(ext.size());
  HIP_TRY(hipMemcpyAsync(pos.data(), d_pos.data(), ext.size() * 4, hipMemcpyDeviceToHost, h.stream));
  h.read_back(&out.out_deg_sum, d_sum.data(), 1);
  for (size_t i = 0; i < pos.size(); ++i) {
    CGA_EXPECTS(pos[i] >= 0, CUGRAPH_INVALID_INPUT, std::string(api) + ": found a source that is not a vertex of the graph");  // bfs.cpp:106-119
    if (pos[i] % t.P == t.rank) out.rows.push_back(pos[i] / t.P);
  }
  return out;
}

paths_result_t* collect(handle_t const& h, mg_traversal_run_t& r, mg_traversal_part_t const& t, bool with_pred, cugraph_data_type_id_t dist_type)
{
  auto ids   = std::make_unique<device_array_t>((size_t)t.n_rows, INT32);
  auto dist  = std::make_unique<device_array_t>((size_t)t.n_rows, dist_type);
  auto preds = std::make_unique<device_array_t>(with_pred ? (size_t)t.n_rows : 0, INT32);
  if (t.n_rows > 0) {
    HIP_TRY(hipMemcpyAsync(ids->buf.ptr, t.local_vertices.data(), (size_t)t.n_rows * 4, hipMemcpyDeviceToDevice, h.stream));
    cugraph_error_t* err = nullptr;
    ck(cugraph_amd_traversal_mg_plan_results(r.plan, dist->buf.ptr, with_pred ? preds->buf.as<int32_t>() : nullptr, &err), err, "results");
  }
  h.sync();
  r.c->check("multi-GPU traversal result");
  return new paths_result_t{ids.release(), dist.release(), preds.release()};
}

}  // namespace

paths_result_t* mg_run_bfs(handle_t& h, graph_t& g, device_array_view_t const* sources, bool direction_optimizing, size_t depth_limit, bool with_pred)
{
  HIP_TRY(hipSetDevice(h.device));
  CGA_EXPECTS(handle_comm(h) == g.mg->comm, CUGRAPH_INVALID_HANDLE, "multi-GPU BFS: the handle is not on the communicator the graph was created on");
  mg_agree(g, [&] {  // rank-local argument checks: every rank fails or none does
    CGA_EXPECTS(sources == nullptr || sources->type == INT32, CUGRAPH_INVALID_INPUT, "vertex type of graph and sources must match");
    if (direction_optimizing)  // bfs_impl.cuh:202-204
      CGA_EXPECTS(g.props.is_symmetric == TRUE, CUGRAPH_INVALID_INPUT, "Invalid input argument: input graph should be symmetric for direction optimizing BFS.");
  }, "cugraph_bfs");
  {
    uint64_t const scalars[3] = {direction_optimizing ? 1ull : 0ull, (uint64_t)depth_limit, with_pred ? 1ull : 0ull};
    mg_agree_same(g, scalars, sizeof(scalars), "cugraph_bfs (direction_optimizing, depth_limit, compute_predecessors)");
  }
  comm_t& c = *g.mg->comm;
  mg_traversal_part_t& t = mg_traversal_part(h, g, false);
  mg_traversal_run_t& r  = ensure_run(h, g, t, 0);
  int const P = t.P, me = t.rank;
  cugraph_error_t* err = nullptr;
  // bottom-up levels (the direction never changes distances or the minimum-external-id parents): needs the in-edge copy
  static bool const allow_bu = !(getenv("CUGRAPH_AMD_MG_BFS_BOTTOM_UP") && std::string(getenv("CUGRAPH_AMD_MG_BFS_BOTTOM_UP")) == "0");
  if (allow_bu && !r.bottom_up_set) {
    mg_traversal_in_edges(h, g, t);
    ck(cugraph_amd_traversal_mg_plan_set_bottom_up(r.plan, t.in_offsets.data(), t.in_indices.data(), t.ext_of_g.data(), &err), err, "set_bottom_up");
    r.bottom_up_set = true;
  }
  bool const bu_ok = r.bottom_up_set;
  std::vector<int32_t> const ext = gather_sources(h, c, sources);
  located_t const loc            = locate_sources(h, g, t, ext, "cugraph_bfs");
  dvec<int32_t> d_rows(std::max<size_t>(loc.rows.size(), 1));
  if (!loc.rows.empty()) HIP_TRY(hipMemcpyAsync(d_rows.data(), loc.rows.data(), loc.rows.size() * 4, hipMemcpyHostToDevice, h.stream));
  h.sync();
  ck(cugraph_amd_traversal_mg_plan_reset(r.plan, loc.rows.empty() ? nullptr : d_rows.data(), loc.rows.size(), (double)FLT_MAX, with_pred ? TRUE : FALSE, &err), err, "reset");
  (void)share_frontier(h, r, t, 0, level_stats_t{0, 0, 0});
  // Beamer's direction rule on GLOBAL sums, the single-GPU driver's constants (traversal.hip: run_bfs)
  double const alpha = getenv("CUGRAPH_AMD_BFS_ALPHA") ? atof(getenv("CUGRAPH_AMD_BFS_ALPHA")) : 60.0;
  double const beta  = getenv("CUGRAPH_AMD_BFS_BETA") ? atof(getenv("CUGRAPH_AMD_BFS_BETA")) : 24.0;
  char const* force  = getenv("CUGRAPH_AMD_MG_BFS");  // "bottomup" / "topdown": pin the direction (tests)
  // "0": the count matrix of a top-down level through the host (round 5's exchange: kept as the cross-check of the device-driven one)
  bool const dev_exchange = !(getenv("CUGRAPH_AMD_MG_BFS_DEVICE_EXCHANGE") && std::string(getenv("CUGRAPH_AMD_MG_BFS_DEVICE_EXCHANGE")) == "0");
  unsigned long long n_front = (unsigned long long)loc.n_sources, frontier_out = loc.out_deg_sum, unvisited_in = (unsigned long long)t.ne_global;
  uint64_t const limit = depth_limit > (size_t)INT32_MAX ? (uint64_t)INT32_MAX : (uint64_t)depth_limit;
  bool bottom_up = false;
  uint64_t level = 0;
  uint64_t steps = 0, bu_levels = 0;
  while (n_front > 0) {
    ++level;
    if (level > limit) break;  // depth_limit is compared after incrementing (bfs_impl.cuh:867-868)
    if (bu_ok) {
      if (!bottom_up) bottom_up = (double)frontier_out > (double)unvisited_in / alpha && n_front > 1024;
      else bottom_up = !((double)n_front < (double)t.nv_global / beta);
      if (force && std::string(force) == "bottomup") bottom_up = true;
      if (force && std::string(force) == "topdown") bottom_up = false;
    }
    // the level's kernels are launched, its counters travel to every rank from the device together with the frontier bits, and the ONE host
    // synchronisation of share_frontier brings back everybody's -- this rank's own included (round 5: the plan's exported apply / bottom_up
    // read them back first, a second synchronisation per level)
    if (bottom_up) {
      mg_plan_bottom_up_launch(r.plan, static_cast<uint32_t const*>(r.bwin[(level - 1) & 1]->local), (uint32_t)level);
      ++bu_levels;
    } else {
      if (dev_exchange) {  // no host in the exchange: the level's only synchronisation is share_frontier's
        mg_plan_expand_launch(r.plan);
        exchange_tuples_dev(h, r, t);
        mg_plan_apply_launch(r.plan, static_cast<int32_t const*>(r.twin->local), 0, (uint32_t)level, static_cast<uint32_t const*>(r.cwin->local));
      } else {
        size_t counts[kCommMaxRanks];
        ck(cugraph_amd_traversal_mg_plan_expand(r.plan, counts, &err), err, "expand");
        size_t const got = exchange_tuples(h, r, P, me, counts);
        mg_plan_apply_launch(r.plan, static_cast<int32_t const*>(r.twin->local), got, (uint32_t)level, nullptr);
      }
    }
    unsigned long long own[4] = {0, 0, 0, 0};
    level_stats_t const tot = share_frontier(h, r, t, (int)(level & 1), level_stats_t{0, 0, 0}, static_cast<counters_t const*>(mg_plan_counters(r.plan)), own);
    if (bottom_up) CGA_EXPECTS(own[0] == own[3], CUGRAPH_UNKNOWN_ERROR, "bottom-up level: discoveries and queue length differ");
    mg_plan_adopt_level(r.plan, (size_t)own[0], own[1], own[2]);
    n_front      = tot.n;
    frontier_out = tot.out_sum;
    unvisited_in = unvisited_in > tot.in_sum ? unvisited_in - tot.in_sum : 0;
    ++steps;
  }
  h.last_stats       = cugraph_amd_traversal_stats_t{};
  h.last_stats.steps = steps;
  h.last_stats.edges_inspected = bu_levels;  // (bottom-up levels taken: what the direction tests look at)
  return collect(h, r, t, with_pred, INT32);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// FLOAT64 weights: the plain exchange loop.  The fast engine packs (float distance, parent) into ONE 64-bit word per vertex, which a double
// distance does not fit; double-weight graphs therefore take this frontier Bellman-Ford, exact in double, written for correctness
// (sssp_impl.cuh:169-566 with weight_t = double): every round the rows whose distance dropped relax their out-edges, the candidates
// (row at the owner, distance, external id of the parent: four words) go to the owners' window, the owner lowers the distance (64-bit
// atomic minimum on the bits of a non-negative double), and the parent is the minimum external id among the candidates that attain the
// vertex's distance -- at the end: among all tight in-edges, the same rule as the float engine and the single-GPU path.
namespace {

__global__ void k_d64_count(int32_t const* front, int64_t n, int32_t const* off, int32_t const* idx, double const* w, unsigned long long const* dist, double cutoff, int64_t L, int P,
                            unsigned long long* counts)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t const u = front[i];
    double const du = __longlong_as_double((long long)dist[u]);
    for (int32_t p = off[u]; p < off[u + 1]; ++p)
      if (du + w[p] < cutoff) atomicAdd(&counts[idx[p] / L], 1ull);
  }
}
__global__ void k_d64_scatter(int32_t const* front, int64_t n, int32_t const* off, int32_t const* idx, double const* w, unsigned long long const* dist, double cutoff, int64_t L,
                              int32_t const* local_vertices, unsigned long long* cursor /*[P]: start of every owner's bucket, advanced*/, int32_t* send)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t const u = front[i];
    double const du = __longlong_as_double((long long)dist[u]);
    for (int32_t p = off[u]; p < off[u + 1]; ++p) {
      double const nd = du + w[p];
      if (!(nd < cutoff)) continue;
      int32_t const g = idx[p];
      unsigned long long const at = atomicAdd(&cursor[g / L], 1ull);
      unsigned long long const b  = (unsigned long long)__double_as_longlong(nd);
      int32_t* t = send + 4 * at;
      t[0] = (int32_t)(g % L); t[1] = (int32_t)(uint32_t)b; t[2] = (int32_t)(uint32_t)(b >> 32); t[3] = local_vertices[u];
    }
  }
}
__global__ void k_d64_lower(int32_t const* recv, int64_t n, unsigned long long* dist, uint32_t* improved)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t const* t = recv + 4 * i;
    unsigned long long const b = (unsigned long long)(uint32_t)t[1] | ((unsigned long long)(uint32_t)t[2] << 32);
    if (b < dist[t[0]] && atomicMin(&dist[t[0]], b) > b) improved[t[0]] = 1u;
  }
}
__global__ void k_d64_reset_pred(uint32_t const* improved, int64_t n_rows, int32_t* pred)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n_rows; i += (int64_t)gridDim.x * blockDim.x) if (improved[i]) pred[i] = INT32_MAX;
}
__global__ void k_d64_parents(int32_t const* recv, int64_t n, unsigned long long const* dist, int32_t source_row, int32_t* pred)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t const* t = recv + 4 * i;
    unsigned long long const b = (unsigned long long)(uint32_t)t[1] | ((unsigned long long)(uint32_t)t[2] << 32);
    if (t[0] != source_row && b == dist[t[0]]) atomicMin(&pred[t[0]], t[3]);
  }
}
__global__ void k_d64_next(uint32_t* improved, int64_t n_rows, int32_t* front, unsigned long long* n_front)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n_rows; i += (int64_t)gridDim.x * blockDim.x)
    if (improved[i]) { improved[i] = 0u; front[atomicAdd(n_front, 1ull)] = (int32_t)i; }
}
__global__ void k_d64_results(unsigned long long const* dist, int32_t const* pred, int64_t n_rows, double* out_d, int32_t* out_p)
{
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  for (; i < n_rows; i += (int64_t)gridDim.x * blockDim.x) {
    out_d[i] = __longlong_as_double((long long)dist[i]);  // unreached: DBL_MAX (sssp_impl.cuh: the weight type's maximum)
    if (out_p) out_p[i] = pred[i] == INT32_MAX ? -1 : pred[i];
  }
}

paths_result_t* mg_run_sssp_f64(handle_t& h, graph_t& g, size_t source, double cutoff, bool with_pred)
{
  comm_t& c              = *g.mg->comm;
  mg_traversal_part_t& t = mg_traversal_part(h, g, true);
  int const P = t.P, me = t.rank;
  CGA_EXPECTS(source <= (size_t)INT32_MAX, CUGRAPH_INVALID_INPUT, "cugraph_sssp: source is not a vertex of the graph");
  std::vector<int32_t> const ext{(int32_t)source};
  located_t const loc = locate_sources(h, g, t, ext, "cugraph_sssp");
  size_t const n1     = (size_t)std::max<int64_t>(t.n_rows, 1);
  dvec<unsigned long long> dist(n1), counts((size_t)P), cursor((size_t)P), n_front(1);
  dvec<int32_t> pred(n1), front(n1), send;
  dvec<uint32_t> improved(n1);
  fill_f64(h, reinterpret_cast<double*>(dist.data()), (int64_t)n1, DBL_MAX);
  fill_i32(h, pred.data(), (int64_t)n1, INT32_MAX);
  HIP_TRY(hipMemsetAsync(improved.data(), 0, n1 * 4, h.stream));
  int32_t const source_row = loc.rows.empty() ? -1 : loc.rows[0];
  int64_t nf               = 0;
  if (source_row >= 0) {
    unsigned long long const zero = 0ull;
    HIP_TRY(hipMemcpyAsync(dist.data() + source_row, &zero, 8, hipMemcpyHostToDevice, h.stream));
    HIP_TRY(hipMemcpyAsync(front.data(), &source_row, 4, hipMemcpyHostToDevice, h.stream));
    nf = 1;
  }
  h.sync();
  comm_window_t* win = nullptr;
  size_t win_tuples  = 0;
  int const channel  = 1;  // (build-time channel: every exchange below is followed by a stream synchronisation)
  uint64_t rounds    = 0;
  double const cut   = cutoff;
  for (;;) {
    ++rounds;
    // candidates per owner, buckets, scatter
    HIP_TRY(hipMemsetAsync(counts.data(), 0, (size_t)P * 8, h.stream));
    if (nf > 0)
      hipLaunchKernelGGL(k_d64_count, grid_for(nf, kBlock, 4096), kBlock, 0, h.stream, (int32_t const*)front.data(), nf, (int32_t const*)t.offsets.data(), (int32_t const*)t.indices.data(),
                         (double const*)t.weights64.data(), (unsigned long long const*)dist.data(), cut, t.L, P, counts.data());
    std::vector<unsigned long long> mine(P);
    h.read_back(mine.data(), counts.data(), (size_t)P);
    std::vector<int64_t> row(P), M((size_t)P * P);
    for (int k = 0; k < P; ++k) row[k] = (int64_t)mine[k];
    c.host_allgather(row.data(), (size_t)P * sizeof(int64_t), M.data());
    int64_t any = 0, need = 1, n_send = 0;
    for (int s = 0; s < P; ++s) {
      int64_t in = 0;
      for (int k = 0; k < P; ++k) { any += M[(size_t)s * P + k]; in += M[(size_t)k * P + s]; }
      need = std::max(need, in);
    }
    if (any == 0) break;  // nobody has a candidate left: every distance is final
    for (int k = 0; k < P; ++k) n_send += row[k];
    if ((size_t)need > win_tuples) {  // collective: every rank sees the same count matrix
      if (win) c.window_free(win);
      win_tuples = (size_t)need * 2;
      win        = c.window_create(win_tuples * 16);
    }
    send.resize_discard((size_t)std::max<int64_t>(n_send, 1) * 4);
    std::vector<unsigned long long> start(P);
    int64_t acc = 0;
    for (int k = 0; k < P; ++k) { start[k] = (unsigned long long)acc; acc += row[k]; }
    HIP_TRY(hipMemcpyAsync(cursor.data(), start.data(), (size_t)P * 8, hipMemcpyHostToDevice, h.stream));
    if (nf > 0)
      hipLaunchKernelGGL(k_d64_scatter, grid_for(nf, kBlock, 4096), kBlock, 0, h.stream, (int32_t const*)front.data(), nf, (int32_t const*)t.offsets.data(), (int32_t const*)t.indices.data(),
                         (double const*)t.weights64.data(), (unsigned long long const*)dist.data(), cut, t.L, (int32_t const*)t.local_vertices.data(), cursor.data(), send.data());
    comm_push_desc_t d{};
    int64_t total = 0;
    for (int s = 0; s < P; ++s) total += M[(size_t)s * P + me];
    for (int k = 0; k < P; ++k) {
      int64_t roff = 0;
      for (int s = 0; s < me; ++s) roff += M[(size_t)s * P + k];
      d.dst[k]   = win->at<int32_t>(k) + roff * 4;
      d.src[k]   = send.data() + (int64_t)start[k] * 4;
      d.words[k] = row[k] * 4;
    }
    d.n = P;
    c.push_multi(h.stream, d);
    c.wait(h.stream, channel, c.signal(h.stream, channel));
    // owner side: lower the distances, then the parents of the distances that stand
    int32_t const* recv = static_cast<int32_t const*>(win->local);
    if (total > 0) {
      hipLaunchKernelGGL(k_d64_lower, grid_for(total, kBlock, 4096), kBlock, 0, h.stream, recv, total, dist.data(), improved.data());
      hipLaunchKernelGGL(k_d64_reset_pred, grid_for(t.n_rows, kBlock, 4096), kBlock, 0, h.stream, (uint32_t const*)improved.data(), t.n_rows, pred.data());
      hipLaunchKernelGGL(k_d64_parents, grid_for(total, kBlock, 4096), kBlock, 0, h.stream, recv, total, (unsigned long long const*)dist.data(), source_row, pred.data());
    }
    HIP_TRY(hipMemsetAsync(n_front.data(), 0, 8, h.stream));
    if (t.n_rows > 0) hipLaunchKernelGGL(k_d64_next, grid_for(t.n_rows, kBlock, 4096), kBlock, 0, h.stream, improved.data(), t.n_rows, front.data(), n_front.data());
    unsigned long long got = 0;
    h.read_back(&got, n_front.data(), 1);
    c.check("multi-GPU SSSP (FLOAT64) round");
    nf = (int64_t)got;
    c.host_barrier();  // nobody overwrites a window before every rank has consumed this round's tuples
  }
  if (win) c.window_free(win);
  auto ids   = std::make_unique<device_array_t>((size_t)t.n_rows, INT32);
  auto dst   = std::make_unique<device_array_t>((size_t)t.n_rows, FLOAT64);
  auto preds = std::make_unique<device_array_t>(with_pred ? (size_t)t.n_rows : 0, INT32);
  if (t.n_rows > 0) {
    HIP_TRY(hipMemcpyAsync(ids->buf.ptr, t.local_vertices.data(), (size_t)t.n_rows * 4, hipMemcpyDeviceToDevice, h.stream));
    hipLaunchKernelGGL(k_d64_results, grid_for(t.n_rows, kBlock, 4096), kBlock, 0, h.stream, (unsigned long long const*)dist.data(), (int32_t const*)pred.data(), t.n_rows,
                       dst->buf.as<double>(), with_pred ? preds->buf.as<int32_t>() : (int32_t*)nullptr);
  }
  h.sync();
  h.last_stats       = cugraph_amd_traversal_stats_t{};
  h.last_stats.steps = rounds;
  return new paths_result_t{ids.release(), dst.release(), preds.release()};
}
}  // namespace

paths_result_t* mg_run_sssp(handle_t& h, graph_t& g, size_t source, double cutoff, bool with_pred)
{
  HIP_TRY(hipSetDevice(h.device));
  CGA_EXPECTS(handle_comm(h) == g.mg->comm, CUGRAPH_INVALID_HANDLE, "multi-GPU SSSP: the handle is not on the communicator the graph was created on");
  CGA_EXPECTS(g.has_weights, CUGRAPH_INVALID_INPUT, "cugraph_sssp requires a weighted graph");  // sssp.cpp:72-73,105 (a property of the graph: the same on every rank)
  {
    uint64_t scalars[3] = {(uint64_t)source, 0, with_pred ? 1ull : 0ull};
    std::memcpy(&scalars[1], &cutoff, sizeof(double));
    mg_agree_same(g, scalars, sizeof(scalars), "cugraph_sssp (source, cutoff, compute_predecessors)");
  }
  if (g.weight_type == FLOAT64) return mg_run_sssp_f64(h, g, source, cutoff, with_pred);
  comm_t& c = *g.mg->comm;
  mg_traversal_part_t& t = mg_traversal_part(h, g, true);
  mg_traversal_run_t& r  = ensure_run(h, g, t, 1);
  int const P = t.P, me = t.rank;
  cugraph_error_t* err = nullptr;
  CGA_EXPECTS(source <= (size_t)INT32_MAX, CUGRAPH_INVALID_INPUT, "cugraph_sssp: source is not a vertex of the graph");
  std::vector<int32_t> const ext{(int32_t)source};
  located_t const loc = locate_sources(h, g, t, ext, "cugraph_sssp");
  dvec<int32_t> d_rows(1);
  if (!loc.rows.empty()) HIP_TRY(hipMemcpyAsync(d_rows.data(), loc.rows.data(), 4, hipMemcpyHostToDevice, h.stream));
  h.sync();
  ck(cugraph_amd_traversal_mg_plan_reset(r.plan, loc.rows.empty() ? nullptr : d_rows.data(), loc.rows.size(), cutoff, with_pred ? TRUE : FALSE, &err), err, "reset");
  // near / far windows (sssp_impl.cuh:233-247, 376-561): delta = 32 x average weight / average degree over the whole graph; a row whose distance
  // drops to a value beyond the window waits in its rank's far pile; when the frontier is empty everywhere the window jumps to the smallest
  // distance left in any pile.  OPT-IN (CUGRAPH_AMD_MG_SSSP_WINDOW=1): measured at RMAT-24, integer weights, 1 / 2 / 4 ranks sharing the GPU:
  // 16.2 / 26.0 / 36.2 ms with windows against 14.4 / 25.3 / 35.5 ms with one unbounded window (18 rounds instead of 13; what a round costs is
  // the successful updates of its wide sweeps, not the relaxations the windows save -- the single-GPU finding of DESIGN.md section 3.4 again)
  bool const windows = getenv("CUGRAPH_AMD_MG_SSSP_WINDOW") && atoi(getenv("CUGRAPH_AMD_MG_SSSP_WINDOW")) != 0;
  if (windows && r.delta == 0.0) {
    dvec<double> d_sum(1);
    HIP_TRY(hipMemsetAsync(d_sum.data(), 0, sizeof(double), h.stream));
    if (t.ne_local > 0) hipLaunchKernelGGL(k_sum_f32, grid_for(t.ne_local, kBlock, 1024), kBlock, 0, h.stream, (float const*)t.weights.data(), t.ne_local, d_sum.data());
    double mine_w = 0.0;
    h.read_back(&mine_w, d_sum.data(), 1);
    std::vector<double> all_w(P);
    c.host_allgather(&mine_w, sizeof(mine_w), all_w.data());
    double sum_w = 0.0;
    for (double x : all_w) sum_w += x;
    double const ne_g = (double)std::max<int64_t>(t.ne_global, 1), nv_g = (double)std::max<int64_t>(t.nv_global, 1);
    r.delta = 32.0 * (sum_w / ne_g) / std::max(ne_g / nv_g, 1.0);
    if (char const* e = getenv("CUGRAPH_AMD_SSSP_DELTA_SCALE")) r.delta *= atof(e);
    if (!(r.delta > 0.0) || !std::isfinite(r.delta)) r.delta = 1.0;
  }
  double upper = windows ? r.delta : (double)FLT_MAX;
  if (windows) ck(cugraph_amd_traversal_mg_plan_sssp_set_window(r.plan, upper, &err), err, "set_window");
  uint64_t rounds = 0, n_windows = 1;
  for (;;) {
    for (;;) {  // the rounds of one window
      ++rounds;
      size_t counts[kCommMaxRanks];
      size_t n_next = 0;
      ck(cugraph_amd_traversal_mg_plan_expand(r.plan, counts, &err), err, "expand");
      size_t const got = exchange_tuples(h, r, P, me, counts);
      ck(cugraph_amd_traversal_mg_plan_apply(r.plan, static_cast<int32_t const*>(r.twin->local), got, (uint32_t)rounds, &n_next, &err), err, "apply");
      c.check("multi-GPU SSSP round");
      int64_t mine = (int64_t)n_next;
      std::vector<int64_t> all(P);
      c.host_allgather(&mine, sizeof(mine), all.data());  // the window is done when the global frontier is empty
      int64_t tot = 0;
      for (auto x : all) tot += x;
      if (tot == 0) break;
    }
    if (!windows) break;
    size_t n_far = 0;
    double dmin  = (double)FLT_MAX;
    ck(cugraph_amd_traversal_mg_plan_sssp_far_stats(r.plan, &n_far, &dmin, &err), err, "far_stats");
    struct far_t { double n, dmin; } mine_f{(double)n_far, dmin};
    std::vector<far_t> all_f(P);
    c.host_allgather(&mine_f, sizeof(mine_f), all_f.data());
    double tot_far = 0.0, gmin = (double)FLT_MAX;
    for (auto const& x : all_f) { tot_far += x.n; gmin = std::min(gmin, x.dmin); }
    if (tot_far == 0.0) break;  // nothing waits beyond the window anywhere: every distance is final
    // the next window starts at the multiple of delta at or below the smallest distance left (never below the current bound: traversal.hip run_sssp)
    double k = std::floor(gmin / r.delta);
    while (k > 0.0 && k * r.delta > gmin) k -= 1.0;
    upper = std::max(upper, k * r.delta) + r.delta;
    size_t n_front = 0;
    ck(cugraph_amd_traversal_mg_plan_sssp_advance(r.plan, upper, &n_front, &err), err, "advance");
    ++n_windows;
  }
  h.last_stats       = cugraph_amd_traversal_stats_t{};
  h.last_stats.steps = rounds;
  h.last_stats.edges_inspected = n_windows;  // (windows taken: what the tests of the near / far schedule look at)
  return collect(h, r, t, with_pred, FLOAT32);
}

}  // namespace cga
