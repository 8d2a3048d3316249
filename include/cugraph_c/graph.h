/* Graph object.  Replaces the SG part of cpp/include/cugraph_c/graph.h
 * (impl cpp/src/c_api/graph_sg.cpp:699 cugraph_graph_create_sg, :835 cugraph_graph_create_with_times_sg
 * -- the one pylibcugraph calls, python/pylibcugraph/pylibcugraph/graphs.pyx:282 --,
 * :989 cugraph_graph_create_sg_from_csr, :1097 cugraph_graph_free).
 *
 * Inputs are copied, never mutated (graph_sg.cpp:98-183).  Type rules follow graph_sg.cpp:745-779:
 * src / dst / `vertices` are INT32 or INT64 (mixed widths promote to INT64, :745-754), weights FLOAT32 or FLOAT64, anything
 * else is CUGRAPH_UNSUPPORTED_TYPE_COMBINATION.  INT64 ids and INT32 ids spread over a sparse range are translated to compact
 * 32-bit internal ids at this boundary (csrc/outer_ids.hip) and translated back in every result; limits, named in the error
 * message: fewer than 2^31 distinct vertex ids per graph (edge counts: see DESIGN.md section 6).
 * edge_ids (INT32 / INT64) and edge_type_ids (INT32) are validated (sizes and types as graph_sg.cpp:781-830) and STORED with their
 * edges (they travel through the build's sort permutation); cugraph_decompress_to_edgelist returns them.  No algorithm behind this
 * library reads them.  Combined with drop_self_loops / drop_multi_edges / symmetrize they are refused with CUGRAPH_NOT_IMPLEMENTED (the
 * rewritten edge list has no one-to-one relation to the columns); multi-GPU graphs (cugraph_graph_create_mg on a communicator handle)
 * and edge start / end times are validated and not kept.
 * drop_self_loops / drop_multi_edges / symmetrize are applied in the reference's order before renumbering
 * (graph_sg.cpp:185-248; csrc/edgelist.hip) and do_expensive_check runs the reference's input checks.
 *
 * Unlike the reference, an algorithm that needs the other storage orientation does not rebuild and
 * re-number the graph (cpp/src/c_api/graph.hpp:84-143): both CSR and CSC live under ONE numbering,
 * the second one built lazily on first use. */
#pragma once
#include <cugraph_c/array.h>
#include <cugraph_c/export.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct { int32_t align_; } cugraph_graph_t;
typedef struct {
  bool_t is_symmetric;
  bool_t is_multigraph;
} cugraph_graph_properties_t;

CUGRAPH_EXPORT cugraph_error_code_t cugraph_graph_create_sg(
  const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
  const cugraph_type_erased_device_array_view_t* vertices,
  const cugraph_type_erased_device_array_view_t* src, const cugraph_type_erased_device_array_view_t* dst,
  const cugraph_type_erased_device_array_view_t* weights,
  const cugraph_type_erased_device_array_view_t* edge_ids,
  const cugraph_type_erased_device_array_view_t* edge_type_ids, bool_t store_transposed, bool_t renumber,
  bool_t drop_self_loops, bool_t drop_multi_edges, bool_t symmetrize, bool_t do_expensive_check,
  cugraph_graph_t** graph, cugraph_error_t** error);

CUGRAPH_EXPORT cugraph_error_code_t cugraph_graph_create_with_times_sg(
  const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
  const cugraph_type_erased_device_array_view_t* vertices,
  const cugraph_type_erased_device_array_view_t* src, const cugraph_type_erased_device_array_view_t* dst,
  const cugraph_type_erased_device_array_view_t* weights,
  const cugraph_type_erased_device_array_view_t* edge_ids,
  const cugraph_type_erased_device_array_view_t* edge_type_ids,
  const cugraph_type_erased_device_array_view_t* edge_start_time_ids,
  const cugraph_type_erased_device_array_view_t* edge_end_time_ids, bool_t store_transposed,
  bool_t renumber, bool_t drop_self_loops, bool_t drop_multi_edges, bool_t symmetrize,
  bool_t do_expensive_check, cugraph_graph_t** graph, cugraph_error_t** error);

CUGRAPH_EXPORT cugraph_error_code_t cugraph_graph_create_sg_from_csr(
  const cugraph_resource_handle_t* handle, const cugraph_graph_properties_t* properties,
  const cugraph_type_erased_device_array_view_t* offsets,
  const cugraph_type_erased_device_array_view_t* indices,
  const cugraph_type_erased_device_array_view_t* weights,
  const cugraph_type_erased_device_array_view_t* edge_ids,
  const cugraph_type_erased_device_array_view_t* edge_type_ids, bool_t store_transposed, bool_t renumber,
  bool_t symmetrize, bool_t do_expensive_check, cugraph_graph_t** graph, cugraph_error_t** error);

/* Per-rank collective creation (cpp/include/cugraph_c/graph.h:238-327, impl cpp/src/c_api/graph_mg.cpp:326-560; called by
 * pylibcugraph.MGGraph, graphs.pyx:648): `num_arrays` arrays of views per column, concatenated.  On a handle created with NULL the
 * graph is a single-GPU graph (renumbered as every MG graph is).  On a handle created on the library's communicator
 * (include/cugraph_amd/extensions.h: cugraph_amd_comm_create; one process per GPU) the call is COLLECTIVE: every rank passes its slice of
 * the edges (and, optionally, of the vertex list) of ONE graph partitioned over the ranks; INT32 ids, FLOAT32 / FLOAT64 weights;
 * drop_self_loops / drop_multi_edges / symmetrize act on the whole graph.  cugraph_pagerank / _personalized_pagerank (all optional
 * arguments), cugraph_bfs, cugraph_sssp, cugraph_louvain, cugraph_degrees / _in_degrees / _out_degrees and cugraph_has_vertex run on
 * such a graph; the other entry points answer CUGRAPH_NOT_IMPLEMENTED for it.  edge ids / types / times are validated and not kept. */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_graph_create_mg(
  cugraph_resource_handle_t const* handle, cugraph_graph_properties_t const* properties,
  cugraph_type_erased_device_array_view_t const* const* vertices, cugraph_type_erased_device_array_view_t const* const* src,
  cugraph_type_erased_device_array_view_t const* const* dst, cugraph_type_erased_device_array_view_t const* const* weights,
  cugraph_type_erased_device_array_view_t const* const* edge_ids,
  cugraph_type_erased_device_array_view_t const* const* edge_type_ids, bool_t store_transposed, size_t num_arrays,
  bool_t drop_self_loops, bool_t drop_multi_edges, bool_t symmetrize, bool_t do_expensive_check, cugraph_graph_t** graph,
  cugraph_error_t** error);

CUGRAPH_EXPORT cugraph_error_code_t cugraph_graph_create_with_times_mg(
  cugraph_resource_handle_t const* handle, cugraph_graph_properties_t const* properties,
  cugraph_type_erased_device_array_view_t const* const* vertices, cugraph_type_erased_device_array_view_t const* const* src,
  cugraph_type_erased_device_array_view_t const* const* dst, cugraph_type_erased_device_array_view_t const* const* weights,
  cugraph_type_erased_device_array_view_t const* const* edge_ids,
  cugraph_type_erased_device_array_view_t const* const* edge_type_ids,
  cugraph_type_erased_device_array_view_t const* const* edge_start_time_ids,
  cugraph_type_erased_device_array_view_t const* const* edge_end_time_ids, bool_t store_transposed, size_t num_arrays,
  bool_t drop_self_loops, bool_t drop_multi_edges, bool_t symmetrize, bool_t do_expensive_check, cugraph_graph_t** graph,
  cugraph_error_t** error);

CUGRAPH_EXPORT void cugraph_graph_free(cugraph_graph_t* graph);
#ifdef __cplusplus
}
#endif
