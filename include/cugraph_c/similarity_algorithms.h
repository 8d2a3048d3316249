/* Present because pylibcugraph's .pxd chain includes it from modules on this library's path
 * (_cugraph_c/graph_functions.pxd -> similarity_algorithms.pxd; e.g. degrees.pyx, louvain.pyx, has_vertex.pyx).
 * Replaces only the TYPES of cpp/include/cugraph_c/similarity_algorithms.h:20-40; the similarity algorithms themselves
 * (Jaccard, Sorensen, overlap, cosine) are outside the PageRank / BFS / SSSP scope and are not declared,
 * so a caller fails at compile time rather than at link time. */
#pragma once
#include <cugraph_c/error.h>
#include <cugraph_c/graph.h>
#include <cugraph_c/graph_functions.h>
#include <cugraph_c/resource_handle.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct { int32_t align_; } cugraph_similarity_result_t;
#ifdef __cplusplus
}
#endif
