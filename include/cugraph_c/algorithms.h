/* Umbrella header.  Replaces cpp/include/cugraph_c/algorithms.h:1-28 -- what every reference C test and
 * pylibcugraph/_cugraph_c/algorithms.pxd include.  It pulls in the algorithm families this library implements
 * (PageRank, BFS / SSSP / path extraction, Louvain); the reference's other families
 * (core_, labeling_, sampling_, similarity_, tree_algorithms.h, lookup_src_dst.h) are outside that scope and have no header
 * here, so a caller of those fails at compile time rather than at link time. */
#pragma once
#include <cugraph_c/error.h>
#include <cugraph_c/export.h>
#include <cugraph_c/graph.h>
#include <cugraph_c/resource_handle.h>

#include <cugraph_c/centrality_algorithms.h>
#include <cugraph_c/community_algorithms.h>
#include <cugraph_c/traversal_algorithms.h>
