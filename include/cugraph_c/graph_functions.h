/* cugraph_has_vertex -- the one symbol of cpp/include/cugraph_c/graph_functions.h:108 on this path
 * (impl cpp/src/c_api/graph_functions.cpp:391; called by pylibcugraph bfs.pyx:140).
 * Returns an owning BOOL device array, one byte per queried external vertex id. */
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
#ifdef __cplusplus
}
#endif
