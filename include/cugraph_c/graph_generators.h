/* RMAT generators behind the reference's API.  Replaces cpp/include/cugraph_c/graph_generators.h:16-178
 * (impl cpp/src/c_api/graph_generators.cpp:24-330; algorithm cpp/src/generators/generate_rmat_edgelist.cuh:38-112 incl.
 * clip_and_flip and the Graph500 id scramble, generate_rmat_edgelists: generate_rmat_edgelist.cuh:114-190).
 *
 * Vertex type is INT32 (scale <= 30).  The reference draws its uniforms from raft::random, which is not vendored: its edge
 * streams cannot be reproduced bit for bit anywhere outside RAFT, and its own tests only check sizes and ranges
 * (cpp/tests/c_api/generate_rmat_test.c).  Here a state is (seed, edges drawn so far) of a counter-based generator, so a
 * fresh state with seed s reproduces the benchmark generator of include/cugraph_amd/extensions.h and the CPU oracle. */
#pragma once
#include <cugraph_c/array.h>
#include <cugraph_c/coo.h>
#include <cugraph_c/error.h>
#include <cugraph_c/export.h>
#include <cugraph_c/graph.h>
#include <cugraph_c/random.h>
#include <cugraph_c/resource_handle.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef enum { POWER_LAW = 0, UNIFORM } cugraph_generator_distribution_t;

CUGRAPH_EXPORT cugraph_error_code_t cugraph_generate_rmat_edgelist(const cugraph_resource_handle_t* handle, cugraph_rng_state_t* rng_state,
                                                                   size_t scale, size_t num_edges, double a, double b, double c,
                                                                   bool_t clip_and_flip, bool_t scramble_vertex_ids, cugraph_coo_t** result,
                                                                   cugraph_error_t** error);
/* n_edgelists lists; list i has scale s_i in [min_scale, max_scale] (size_distribution) and edge_factor * 2^s_i edges drawn with
 * (a, b, c) = (0.57, 0.19, 0.19) when edge_distribution = POWER_LAW, (0.25, 0.25, 0.25) when UNIFORM
 * (cpp/src/generators/generate_rmat_edgelist.cuh:114-190) */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_generate_rmat_edgelists(const cugraph_resource_handle_t* handle, cugraph_rng_state_t* rng_state,
                                                                    size_t n_edgelists, size_t min_scale, size_t max_scale, size_t edge_factor,
                                                                    cugraph_generator_distribution_t size_distribution,
                                                                    cugraph_generator_distribution_t edge_distribution, bool_t clip_and_flip,
                                                                    bool_t scramble_vertex_ids, cugraph_coo_list_t** result,
                                                                    cugraph_error_t** error);
/* uniform weights in [minimum_weight, maximum_weight), FLOAT32 or FLOAT64, attached to the coo */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_generate_edge_weights(const cugraph_resource_handle_t* handle, cugraph_rng_state_t* rng_state,
                                                                  cugraph_coo_t* coo, cugraph_data_type_id_t dtype, double minimum_weight,
                                                                  double maximum_weight, cugraph_error_t** error);
/* edge ids 0 .. num_edges - 1 (type of the vertex column), attached to the coo; multi_gpu must be FALSE on a one-rank handle */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_generate_edge_ids(const cugraph_resource_handle_t* handle, cugraph_coo_t* coo, bool_t multi_gpu,
                                                              cugraph_error_t** error);
/* uniform INT32 edge types in [min_edge_type, max_edge_type], attached to the coo */
CUGRAPH_EXPORT cugraph_error_code_t cugraph_generate_edge_types(const cugraph_resource_handle_t* handle, cugraph_rng_state_t* rng_state,
                                                                cugraph_coo_t* coo, int32_t min_edge_type, int32_t max_edge_type,
                                                                cugraph_error_t** error);
#ifdef __cplusplus
}
#endif
