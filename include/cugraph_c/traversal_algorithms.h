/* BFS / SSSP.  Replaces cpp/include/cugraph_c/traversal_algorithms.h:24-165
 * (impl cpp/src/c_api/bfs.cpp:189 cugraph_bfs, :156-187 result accessors; cpp/src/c_api/sssp.cpp:136).
 *
 * BFS distances/predecessors have the graph's vertex type, SSSP distances its weight type; unreached
 * = type max, predecessor -1; the predecessor array has size 0 when not requested.  A source array
 * whose type differs from the graph's vertex type, or a source that is not a vertex of the graph,
 * gives CUGRAPH_INVALID_INPUT (bfs.cpp:106-119, 198-205).  depth_limit semantics bfs_impl.cuh:867-868.
 * Predecessors are deterministic here: BFS returns, among the valid parents (one level up), the one with the smallest
 * INTERNAL id (= the highest-degree one; equal to the smallest external id when the graph is not renumbered); SSSP the
 * lexicographic min (distance, external parent id) as sssp_impl.cuh:334 -- both valid instances of the reference's
 * reduce_op::any, whose own tests accept any valid parent (bfs_test.cpp:217-233).
 * direction_optimizing: TRUE on a non-symmetric graph is CUGRAPH_INVALID_INPUT as in the reference (bfs_impl.cuh:202-204).
 * The flag does not select the algorithm here: levels run bottom-up whenever the in-edges are at hand (symmetric graph,
 * or a CSC built by an earlier call) and Beamer's heuristic says so -- distances do not depend on the direction, and the
 * parent rule above is direction-independent too.  CUGRAPH_AMD_BFS=topdown pins the push-only path.
 *
 * cugraph_extract_paths (traversal_algorithms.h:167-201, impl cpp/src/c_api/extract_paths.cpp:24-170 over
 * cugraph::extract_bfs_paths, cpp/src/traversal/extract_bfs_paths_impl.cuh:130-240): for every destination the path from its
 * BFS source, as one row of a row-major (n_destinations x max_path_length) matrix of external ids padded with -1;
 * max_path_length = 1 + the largest hop count among the destinations.  Takes the result of cugraph_bfs with predecessors. */
#pragma once
#include <cugraph_c/error.h>
#include <cugraph_c/graph.h>
#include <cugraph_c/resource_handle.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct { int32_t align_; } cugraph_paths_result_t;
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_paths_result_get_vertices(cugraph_paths_result_t* result);
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_paths_result_get_distances(cugraph_paths_result_t* result);
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_paths_result_get_predecessors(cugraph_paths_result_t* result);
CUGRAPH_EXPORT void cugraph_paths_result_free(cugraph_paths_result_t* result);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_bfs(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph,
  cugraph_type_erased_device_array_view_t* sources, bool_t direction_optimizing, size_t depth_limit,
  bool_t compute_predecessors, bool_t do_expensive_check, cugraph_paths_result_t** result,
  cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_error_code_t cugraph_sssp(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph, size_t source, double cutoff,
  bool_t compute_predecessors, bool_t do_expensive_check, cugraph_paths_result_t** result,
  cugraph_error_t** error);
typedef struct { int32_t align_; } cugraph_extract_paths_result_t;
CUGRAPH_EXPORT cugraph_error_code_t cugraph_extract_paths(
  const cugraph_resource_handle_t* handle, cugraph_graph_t* graph, const cugraph_type_erased_device_array_view_t* sources,
  const cugraph_paths_result_t* paths_result, const cugraph_type_erased_device_array_view_t* destinations,
  cugraph_extract_paths_result_t** result, cugraph_error_t** error);
CUGRAPH_EXPORT size_t cugraph_extract_paths_result_get_max_path_length(cugraph_extract_paths_result_t* result);
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_extract_paths_result_get_paths(cugraph_extract_paths_result_t* result);
CUGRAPH_EXPORT void cugraph_extract_paths_result_free(cugraph_extract_paths_result_t* result);
#ifdef __cplusplus
}
#endif
