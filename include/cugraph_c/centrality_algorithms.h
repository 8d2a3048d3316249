/* PageRank family.  Replaces the pagerank part of cpp/include/cugraph_c/centrality_algorithms.h:24-330
 * (impl cpp/src/c_api/pagerank.cpp:247 cugraph_pagerank, :316 _allow_nonconvergence,
 * :378 cugraph_personalized_pagerank, :466 _allow_nonconvergence; result accessors
 * cpp/src/c_api/centrality_result.cpp).
 *
 * Optional (vertices, values) pairs carry EXTERNAL vertex ids.  The result's vertex column is the
 * graph's number_map (internal order), values are of the graph's weight type.  The non-`allow`
 * variants return CUGRAPH_UNKNOWN_ERROR "PageRank failed to converge." while still setting *result
 * (pagerank.cpp:306-313).  Iteration order and stopping rule: pagerank_impl.cuh:224-329. */
#pragma once
#include <cugraph_c/array.h>
#include <cugraph_c/error.h>
#include <cugraph_c/graph.h>
#include <cugraph_c/resource_handle.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct { int32_t align_; } cugraph_centrality_result_t;

CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_centrality_result_get_vertices(
  cugraph_centrality_result_t* result);
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_centrality_result_get_values(
  cugraph_centrality_result_t* result);
CUGRAPH_EXPORT size_t cugraph_centrality_result_get_num_iterations(cugraph_centrality_result_t* result);
CUGRAPH_EXPORT bool_t cugraph_centrality_result_converged(cugraph_centrality_result_t* result);
CUGRAPH_EXPORT void cugraph_centrality_result_free(cugraph_centrality_result_t* result);

#define CUGRAPH_PAGERANK_COMMON_ARGS                                                              \
  const cugraph_resource_handle_t *handle, cugraph_graph_t *graph,                                \
    const cugraph_type_erased_device_array_view_t *precomputed_vertex_out_weight_vertices,       \
    const cugraph_type_erased_device_array_view_t *precomputed_vertex_out_weight_sums,           \
    const cugraph_type_erased_device_array_view_t *initial_guess_vertices,                       \
    const cugraph_type_erased_device_array_view_t *initial_guess_values

CUGRAPH_EXPORT cugraph_error_code_t cugraph_pagerank(
  CUGRAPH_PAGERANK_COMMON_ARGS, double alpha, double epsilon, size_t max_iterations,
  bool_t do_expensive_check, cugraph_centrality_result_t** result, cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_pagerank_allow_nonconvergence(
  CUGRAPH_PAGERANK_COMMON_ARGS, double alpha, double epsilon, size_t max_iterations,
  bool_t do_expensive_check, cugraph_centrality_result_t** result, cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_personalized_pagerank(
  CUGRAPH_PAGERANK_COMMON_ARGS, const cugraph_type_erased_device_array_view_t* personalization_vertices,
  const cugraph_type_erased_device_array_view_t* personalization_values, double alpha, double epsilon,
  size_t max_iterations, bool_t do_expensive_check, cugraph_centrality_result_t** result,
  cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_personalized_pagerank_allow_nonconvergence(
  CUGRAPH_PAGERANK_COMMON_ARGS, const cugraph_type_erased_device_array_view_t* personalization_vertices,
  const cugraph_type_erased_device_array_view_t* personalization_values, double alpha, double epsilon,
  size_t max_iterations, bool_t do_expensive_check, cugraph_centrality_result_t** result,
  cugraph_error_t** error);
#ifdef __cplusplus
}
#endif
