// The multi-GPU graph behind cugraph_graph_create_mg on a communicator handle (replaces cpp/src/c_api/graph_mg.cpp:140-560 and the
// multi_gpu = true halves of cpp/src/structure/create_graph_from_edgelist_impl.cuh:473-955 / renumber_edgelist_impl.cuh:425-829).
//
// Every rank hands over ITS slice of the edge list (external ids).  The graph object keeps that slice and builds, on the first call
// that needs it, the partition an algorithm family runs on -- both deal the vertices of a global degree order round-robin over the
// P ranks (position p -> rank p % P, local row p / P: every rank gets the same hub / tail mix):
//   * PageRank: 1-D by DESTINATION in descending global in-degree order (the owner holds all in-edges of its rows: no partial-sum
//     reduction); columns are numbered so that the receive window of the per-iteration x exchange IS the gather vector
//     (grouped by the owning peer, hottest first inside a group: no unpack pass);
//   * BFS / SSSP: 1-D by SOURCE in descending global out-degree order, destinations as compact global ids g = owner * L + row.
// The reference hashes vertices to ranks and keeps a 2-D edge partition (partition_manager.hpp:42-51); DESIGN.md section 5 has the
// byte counts that decide for 1-D on one xGMI node.
#pragma once

#include <functional>

#include "common.hpp"

namespace cga {

struct comm_t;
struct comm_window_t;
struct mg_traversal_run_t;  // traversal_mg_driver.hip: the per-rank plan + exchange windows of one traversal family
struct paths_result_t;

struct mg_pagerank_part_t {
  int P{1}, rank{0};
  int64_t nv_global{0}, n_rows{0}, ncols{0}, n_send{0}, ne_local{0};
  dvec<int32_t> local_vertices;      // [n_rows] external id of local row r
  dvec<int32_t> send_index;          // [n_send] local row whose x is the k-th value this rank sends, grouped by receiving peer
  std::vector<int64_t> send_first;   // [P + 1] k-range per receiving peer
  std::vector<int64_t> dst_off;      // [P] element offset in peer r's x window where this rank's values for r start
  std::vector<int64_t> seg_start;    // [P + 1] layout of this rank's own window: values of owner s at [seg_start[s], seg_start[s + 1])
  dev_buf outw_local;                // [n_rows] out-weight sums of the owned vertices (weight type)
  cugraph_graph_t* local{nullptr};   // CSC of the owned rows over the window columns (renumber = FALSE); owned
  ~mg_pagerank_part_t();
};

// The reference's 2-D layout (graph_view.hpp:64-230, partition_manager.hpp:40-51) behind the same entry points: P = R x C ranks, rank = c * R + r;
// vertex partition q = position % P of the global descending in-degree order, L = ceil(V / P) rows each; rank (r, c) OWNS partition c * R + r and
// STORES the edges whose source lies in the partitions [c * R, (c + 1) * R) (local column (q_src % R) * L + row) and whose destination lies in the
// partitions {i * R + r} (local row (q_dst / R) * L + row).  PageRank: all-gather of x over the COLUMN group {c * R + r'}, SpMV of the block,
// reduce of the partial rows over the ROW group {c' * R + r} to their owners.  Chosen with CUGRAPH_AMD_MG_LAYOUT=2d (every rank the same).
struct mg_pagerank2d_part_t {
  int P{1}, R{1}, C{1}, r{0}, c{0}, rank{0};
  int64_t nv_global{0}, L{0}, n_own{0}, ne_local{0};
  dvec<int32_t> local_vertices;      // [n_own] external ids of the owned rows
  dev_buf outw_own;                  // [L] out-weight sums of the owned vertices (weight type; padded rows 0)
  cugraph_graph_t* local{nullptr};   // the block: CSC over max(C, R) * L ids, rows = local rows, columns = local columns (renumber = FALSE); owned
  ~mg_pagerank2d_part_t();
};

struct mg_traversal_part_t {
  int P{1}, rank{0};
  int64_t nv_global{0}, n_rows{0}, L{0}, ne_local{0}, ne_global{0};
  dvec<int32_t> local_vertices;  // [max(n_rows, 1)] external ids
  dvec<int32_t> offsets, indices;  // CSR of the owned rows' out-edges, destinations as compact global ids, ascending inside a row
  dvec<float> weights;             // optional (FLOAT32 graphs)
  dvec<double> weights64;          // optional (FLOAT64 graphs: the plain exchange loop of traversal_mg_driver.hip)
  bool has_weights{false};
  dvec<int32_t> pos;               // [vrange] external id - vmin -> position in the out-degree order (-1: not a vertex)
  dvec<uint32_t> out_deg;          // [vrange] global out-degrees (the direction heuristic needs the sources' sum)
  // bottom-up BFS levels: in-edges of the owned rows, neighbours as compact global ids in ascending order of their external id
  bool has_in{false};
  dvec<int32_t> in_offsets, in_indices, ext_of_g;
  std::shared_ptr<mg_traversal_run_t> run;  // created by the first traversal, freed with the graph (collective)
};

struct mg_graph_t {
  comm_t* comm{nullptr};
  edge_list_t el;             // this rank's slice (external ids), after the local part of the creation flags
  dev_buf edge_ids;           // optional edge ids of the slice, in slice order (ids_size bytes each; graph_mg.cpp:127-151 keeps them as edge properties);
  size_t ids_size{0};         // no algorithm of this library reads them: they come back from cugraph_decompress_to_edgelist
  dvec<int32_t> edge_types;   // optional edge type ids of the slice
  bool has_edge_types{false};
  dvec<int32_t> listed;       // vertices this rank listed explicitly (isolated vertices exist only through such a list)
  int64_t n_listed{0};
  int64_t vmin{0}, vrange{0};  // dense external id range over all ranks
  int64_t nv_global{0}, ne_global{0};
  dvec<uint32_t> present;     // [vrange] 1 = the id is a vertex of the graph
  std::unique_ptr<mg_pagerank_part_t> pr;
  std::unique_ptr<mg_pagerank2d_part_t> pr2d;
  std::unique_ptr<mg_traversal_part_t> tr[2];  // [0] = without weights (BFS), [1] = with the graph's weights (SSSP)
};

struct mg_column_t {
  void const* ptr;
  size_t elem;  // 4 or 8 bytes
};
// all-to-all-v of whole columns: element i of every column goes to rank owner[i]; returns the received columns (grouped by sender)
int64_t mg_shuffle_by_owner(handle_t const& h, comm_t& c, int32_t const* owner, int64_t m, std::vector<mg_column_t> const& cols, std::vector<dev_buf>& out);
struct clustering_result_t;
// cugraph_louvain on a multi-GPU graph (louvain.hip): collective; every rank gets the clusters of the vertices it owns (v % P == rank in ascending id order)
clustering_result_t* mg_run_louvain(handle_t const& h, graph_t& g, size_t max_level, double threshold, double resolution);

// INT64 vertex ids on a multi-GPU graph (collective; graph_mg.cpp:127-151 instantiates vertex_t = int64_t): when ANY rank hands over an INT64
// column, every rank gets the same ascending list of the distinct ids of all ranks in `outer` (outer_ids.hip: compact id = position) and
// returns true; the caller then replaces its columns by compact int32 ids and results leave through outer_replace_ids
bool mg_outer_ids(handle_t const& h, device_array_view_t const* vertices, device_array_view_t const* src, device_array_view_t const* dst, outer_ids_t& outer);
// cugraph_graph_create_mg / _with_times_mg on a handle with more than one rank (collective)
void mg_graph_create(handle_t const& h, graph_t& g, device_array_view_t const* vertices, device_array_view_t const* src, device_array_view_t const* dst,
                     device_array_view_t const* weights, device_array_view_t const* edge_ids, device_array_view_t const* edge_type_ids, bool drop_self_loops,
                     bool drop_multi_edges, bool symmetrize);
// cugraph_extract_paths on a multi-GPU graph (collective): every rank's (vertex, hop count, predecessor) triples -- external int32 ids -- folded into
// two tables over the dense id range that every rank holds afterwards: dist1[id - vmin] = hop count + 1, pred2[id - vmin] = predecessor + 2
// (0 = no rank reported the id)
void mg_gather_paths(handle_t const& h, graph_t& g, int32_t const* vertices, int32_t const* dist, int32_t const* pred, int64_t n, dvec<uint32_t>& dist1,
                     dvec<uint32_t>& pred2);
// max over the ranks of a host value (collective)
int64_t mg_host_max(graph_t& g, int64_t mine);
// Collective: the rank-local argument checks of a collective entry point.  Every rank runs `local_checks` (which throws api_error on a bad argument),
// the verdicts are exchanged, and if ANY rank failed EVERY rank throws -- the failing ranks their own error, the others CUGRAPH_INVALID_INPUT naming the
// first failing rank -- so that nobody enters the algorithm's first collective alone (a lone rank used to leave its peers in a barrier until the
// communicator's timeout and poison the session).
void mg_agree(graph_t& g, std::function<void()> const& local_checks, char const* what);
// Collective: every rank must pass the same `bytes` (<= 64) of scalar arguments; CUGRAPH_INVALID_INPUT on every rank otherwise
void mg_agree_same(graph_t& g, void const* blob, size_t bytes, char const* what);
void mg_has_vertex(handle_t const& h, graph_t const& g, int32_t const* v, int64_t n, uint8_t* out);
// cugraph_degrees family on a multi-GPU graph (collective): this rank's share of the vertices (all, or those any rank listed) and their degrees
int64_t mg_degrees(handle_t const& h, graph_t& g, device_array_view_t const* listed, bool want_in, bool want_out, dvec<int32_t>& ids, dvec<int32_t>& in_deg, dvec<int32_t>& out_deg);
mg_pagerank_part_t& mg_pagerank_part(handle_t const& h, graph_t& g);                   // collective on first use
mg_pagerank2d_part_t& mg_pagerank2d_part(handle_t const& h, graph_t& g);               // collective on first use
void mg_grid_shape(int P, int* R, int* C);  // R = the largest divisor of P that is <= sqrt(P) (cpp/tests/utilities/mg_utilities.cpp:48-52): 1x2, 2x2, 2x4 for 2, 4, 8 ranks
mg_traversal_part_t& mg_traversal_part(handle_t const& h, graph_t& g, bool weighted);  // collective on first use
void mg_traversal_in_edges(handle_t const& h, graph_t& g, mg_traversal_part_t& t);     // collective: the in-edge copy for bottom-up BFS levels
// cugraph_bfs / cugraph_sssp on a multi-GPU graph (traversal_mg_driver.hip): collective; every rank gets its owned vertices back
paths_result_t* mg_run_bfs(handle_t& h, graph_t& g, device_array_view_t const* sources, bool direction_optimizing, size_t depth_limit, bool with_pred);
paths_result_t* mg_run_sssp(handle_t& h, graph_t& g, size_t source, double cutoff, bool with_pred);

}  // namespace cga
