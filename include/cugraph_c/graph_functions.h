/* cugraph_has_vertex (cpp/include/cugraph_c/graph_functions.h:108, impl cpp/src/c_api/graph_functions.cpp:391; called by
 * pylibcugraph bfs.pyx:140): returns an owning BOOL device array, one byte per queried external vertex id.
 * cugraph_degrees / cugraph_in_degrees / cugraph_out_degrees + result accessors (graph_functions.h:285-393, impl
 * cpp/src/c_api/degrees.cu:24-213, degrees_result.cpp:9-57): source_vertices = NULL means every vertex (result vertex order =
 * internal order); a result accessor returns NULL for a side that was not requested, except that a symmetric graph's
 * out-degrees are served from its in-degrees.  Each accessor call returns a new heap view. */
#pragma once
#include <cugraph_c/error.h>
#include <cugraph_c/graph.h>
#include <cugraph_c/resource_handle.h>
#ifdef __cplusplus
extern "C" {
#endif
CUGRAPH_EXPORT cugraph_error_code_t cugraph_has_vertex(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
  cugraph_type_erased_device_array_view_t* vertices, bool_t do_expensive_check,
  cugraph_type_erased_device_array_t** result, cugraph_error_t** error);
typedef struct { int32_t align_; } cugraph_degrees_result_t;
CUGRAPH_EXPORT cugraph_error_code_t cugraph_in_degrees(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
                                                       const cugraph_type_erased_device_array_view_t* source_vertices,
                                                       bool_t do_expensive_check, cugraph_degrees_result_t** result,
                                                       cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_out_degrees(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
                                                        const cugraph_type_erased_device_array_view_t* source_vertices,
                                                        bool_t do_expensive_check, cugraph_degrees_result_t** result,
                                                        cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_degrees(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
                                                    const cugraph_type_erased_device_array_view_t* source_vertices,
                                                    bool_t do_expensive_check, cugraph_degrees_result_t** result,
                                                    cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_degrees_result_get_vertices(cugraph_degrees_result_t* degrees_result);
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_degrees_result_get_in_degrees(cugraph_degrees_result_t* degrees_result);
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_degrees_result_get_out_degrees(cugraph_degrees_result_t* degrees_result);
CUGRAPH_EXPORT void cugraph_degrees_result_free(cugraph_degrees_result_t* degrees_result);

/* cugraph_decompress_to_edgelist (graph_functions.h:399-480, impl cpp/src/c_api/decompress_to_edgelist.cpp:24-125; called by
 * pylibcugraph decompress_to_edgelist.pyx): the graph as (sources, destinations[, weights]) with external ids, in by-source
 * storage order.  Edge ids / types are not stored by this library: those accessors and the offsets accessor return NULL. */
typedef struct { int32_t align_; } cugraph_edgelist_t;
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_edgelist_get_sources(cugraph_edgelist_t* edgelist);
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_edgelist_get_destinations(cugraph_edgelist_t* edgelist);
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_edgelist_get_edge_weights(cugraph_edgelist_t* edgelist);
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_edgelist_get_edge_ids(cugraph_edgelist_t* edgelist);
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_edgelist_get_edge_type_ids(cugraph_edgelist_t* edgelist);
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_edgelist_get_edge_offsets(cugraph_edgelist_t* edgelist);
CUGRAPH_EXPORT void cugraph_edgelist_free(cugraph_edgelist_t* edgelist);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_decompress_to_edgelist(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
                                                                   bool_t do_expensive_check, cugraph_edgelist_t** result,
                                                                   cugraph_error_t** error);

/* opaque types the reference declares in this header and that pylibcugraph's .pxd chain names
 * (_cugraph_c/graph_functions.pxd); the functions that produce them (vertex pairs, induced subgraphs) are outside the
 * PageRank / BFS / SSSP scope of this library and are not declared */
typedef struct { int32_t align_; } cugraph_vertex_pairs_t;
typedef struct { int32_t align_; } cugraph_induced_subgraph_result_t;
#ifdef __cplusplus
}
#endif
