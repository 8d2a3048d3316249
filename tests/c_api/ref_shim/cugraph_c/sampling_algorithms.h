/* TEST INFRASTRUCTURE, not part of include/.  The reference's cpp/tests/c_api/c_test_utils.h:9,126-145 includes this header
 * and names two of its types in the prototype of validate_sample_result(); create_graph_test.c:253-433 (the one test of that
 * file this library does not run) calls a handful of sampling entry points.  Sampling is outside the PageRank / BFS / SSSP
 * scope, so this stand-in declares only what those two files need to COMPILE; the definitions in
 * ref_test_shim.c return CUGRAPH_NOT_IMPLEMENTED.  Names and signatures: cpp/include/cugraph_c/sampling_algorithms.h. */
#pragma once
#include <cugraph_c/error.h>
#include <cugraph_c/graph.h>
#include <cugraph_c/random.h>
#include <cugraph_c/resource_handle.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct { int32_t align_; } cugraph_sample_result_t;
typedef struct { int32_t align_; } cugraph_sampling_options_t;
typedef enum cugraph_prior_sources_behavior_t { DEFAULT = 0, CARRY_OVER, EXCLUDE } cugraph_prior_sources_behavior_t;
typedef enum cugraph_compression_type_t { COO = 0, CSR, CSC, DCSR, DCSC } cugraph_compression_type_t;
cugraph_error_code_t cugraph_sampling_options_create(cugraph_sampling_options_t** options, cugraph_error_t** error);
void cugraph_sampling_set_renumber_results(cugraph_sampling_options_t* options, bool_t value);
void cugraph_sampling_set_compress_per_hop(cugraph_sampling_options_t* options, bool_t value);
void cugraph_sampling_set_with_replacement(cugraph_sampling_options_t* options, bool_t value);
void cugraph_sampling_set_return_hops(cugraph_sampling_options_t* options, bool_t value);
void cugraph_sampling_set_compression_type(cugraph_sampling_options_t* options, cugraph_compression_type_t value);
void cugraph_sampling_set_prior_sources_behavior(cugraph_sampling_options_t* options, cugraph_prior_sources_behavior_t value);
void cugraph_sampling_set_dedupe_sources(cugraph_sampling_options_t* options, bool_t value);
void cugraph_sampling_options_free(cugraph_sampling_options_t* options);
cugraph_error_code_t cugraph_homogeneous_uniform_neighbor_sample(
  const cugraph_resource_handle_t* handle, cugraph_rng_state_t* rng_state, cugraph_graph_t* graph,
  const cugraph_type_erased_device_array_view_t* start_vertices, const cugraph_type_erased_device_array_view_t* starting_vertex_label_offsets,
  const cugraph_type_erased_host_array_view_t* fan_out, const cugraph_sampling_options_t* options, bool_t do_expensive_check,
  cugraph_sample_result_t** result, cugraph_error_t** error);
void cugraph_sample_result_free(cugraph_sample_result_t* result);
#ifdef __cplusplus
}
#endif
