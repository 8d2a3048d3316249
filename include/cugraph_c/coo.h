/* Edge lists in coordinate form as the generators return them.  Replaces cpp/include/cugraph_c/coo.h:16-110
 * (impl cpp/src/c_api/graph_generators.cpp:24-120).  cugraph_coo_get_edge_id / _edge_type return NULL until
 * cugraph_generate_edge_ids / cugraph_generate_edge_types attached a column (as in the reference). */
#pragma once
#include <cugraph_c/array.h>
#include <cugraph_c/export.h>
#include <cugraph_c/graph.h>
#include <cugraph_c/random.h>
#include <cugraph_c/resource_handle.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct { int32_t align_; } cugraph_coo_t;
typedef struct { int32_t align_; } cugraph_coo_list_t;
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_coo_get_sources(cugraph_coo_t* coo);
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_coo_get_destinations(cugraph_coo_t* coo);
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_coo_get_edge_weights(cugraph_coo_t* coo);
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_coo_get_edge_id(cugraph_coo_t* coo);
CUGRAPH_EXPORT cugraph_type_erased_device_array_view_t* cugraph_coo_get_edge_type(cugraph_coo_t* coo);
CUGRAPH_EXPORT size_t cugraph_coo_list_size(const cugraph_coo_list_t* coo_list);
CUGRAPH_EXPORT cugraph_coo_t* cugraph_coo_list_element(cugraph_coo_list_t* coo_list, size_t index);
CUGRAPH_EXPORT void cugraph_coo_free(cugraph_coo_t* coo);
CUGRAPH_EXPORT void cugraph_coo_list_free(cugraph_coo_list_t* coo_list);
#ifdef __cplusplus
}
#endif
