/* Louvain.  Replaces cpp/include/cugraph_c/community_algorithms.h:64-176 for cugraph_louvain and the hierarchical clustering
 * result (impl cpp/src/c_api/louvain.cpp:24-135, hierarchical_clustering_result.cpp; algorithm cpp/src/community/louvain_impl.cuh:40-287,
 * detail/common_methods.cuh:52-479).
 *
 * max_level bounds the number of contraction levels, threshold is the minimum modularity gain of a sweep (and, divided by the
 * number of vertices, of a single move), resolution the gamma of the modularity formula.  An unweighted graph is treated as
 * weight 1 per edge (louvain.cpp:86-92).  The graph must list both directions of an undirected edge (is_symmetric graphs do).
 * The run is deterministic (the C API passes no random state).  Cluster ids are dense, 0 .. n_clusters - 1; which cluster gets
 * which id is an artefact of the contraction's renumbering in the reference -- here: the rank of the cluster's label -- and
 * the reference's two C-API goldens (cpp/tests/c_api/louvain_test.c) come out with identical ids.
 * Other entry points of the reference header (Leiden, ECG, triangle count, ...) are outside SURVEY section 8. */
#pragma once
#include <cugraph_c/error.h>
#include <cugraph_c/graph.h>
#include <cugraph_c/resource_handle.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct { int32_t align_; } cugraph_hierarchical_clustering_result_t;
CUGRAPH_EXPORT cugraph_error_code_t cugraph_louvain(const cugraph_resource_handle_t* handle, cugraph_graph_t* graph, size_t max_level,
                                                    double threshold, double resolution, bool_t do_expensive_check,
                                                    cugraph_hierarchical_clustering_result_t** result, cugraph_error_t** error);
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_hierarchical_clustering_result_get_vertices(
  cugraph_hierarchical_clustering_result_t* result);
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_hierarchical_clustering_result_get_clusters(
  cugraph_hierarchical_clustering_result_t* result);
CUGRAPH_EXPORT double cugraph_hierarchical_clustering_result_get_modularity(cugraph_hierarchical_clustering_result_t* result);
CUGRAPH_EXPORT void cugraph_hierarchical_clustering_result_free(cugraph_hierarchical_clustering_result_t* result);
#ifdef __cplusplus
}
#endif
