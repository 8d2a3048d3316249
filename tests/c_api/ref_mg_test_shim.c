/* TEST INFRASTRUCTURE: the helper functions the reference's plain-C MULTI-GPU tests expect from its test library (declared in
 * cpp/tests/c_api/mg_test_utils.h:52-161, defined there in mg_test_utils.cpp on top of MPI + NCCL + raft), written against
 * include/cugraph_c + include/cugraph_amd/extensions.h so that cpp/tests/c_api/mg_{pagerank,bfs,sssp,louvain,create_graph}_test.c
 * compile UNCHANGED, in place, and run against this library on N ranks (tests/c_api/build_ref_tests.sh, tests/test_reference_c_tests.py).
 *
 * Where the reference is started by mpirun and create_mg_raft_handle() calls MPI_Init, this create_mg_raft_handle() FORKS the ranks
 * (CUGRAPH_AMD_TEST_RANKS, default 2; before anything touches the GPU) and every process -- rank 0 = the original one -- continues as one
 * rank on the library's own communicator (cugraph_amd_comm_create), whose address stands where the reference passes a raft::handle_t* into
 * cugraph_create_resource_handle.  Behaviour mirrored: graphs are created from edges that live on rank 0 only (mg_test_utils.cpp:150-257:
 * "COO is assumed to be defined entirely on rank 0"), is_multigraph = TRUE, cugraph_graph_create_with_times_mg with one edge list;
 * run_mg_test() sums the ranks' results so that every rank reports a failure anywhere (mg_test_utils.cpp:54-101).
 * All ranks share HIP device 0 unless CUGRAPH_AMD_TEST_DEVICES=n spreads them (HIP_VISIBLE_DEVICES = rank % n). */
#define _POSIX_C_SOURCE 200809L
#include "mg_test_utils.h" /* the reference's own header (resolved through -I to cpp/tests/c_api; <mpi.h> = tests/c_api/ref_shim/mpi.h) */

#include <cugraph_amd/extensions.h>
#include <cugraph_c/array.h>

#include <math.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

static cugraph_amd_comm_t* g_comm = NULL;
static int g_rank = 0, g_size = 1;
static pid_t g_children[64];

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

/* sum of one value per rank (host side, through the communicator's bootstrap segment) */
static long long sum_over_ranks(long long v)
{
  long long all[64];
  cugraph_error_t* err = NULL;
  if (g_size == 1) return v;
  if (cugraph_amd_comm_host_allgather(g_comm, &v, sizeof(v), all, &err) != CUGRAPH_SUCCESS) {
    printf("rank %d: host all-gather failed: %s\n", g_rank, err ? cugraph_error_message(err) : "?");
    exit(3);
  }
  long long s = 0;
  for (int r = 0; r < g_size; ++r) s += all[r];
  return s;
}

size_t cugraph_size_t_allreduce(const cugraph_resource_handle_t* handle, size_t value)
{
  (void)handle;
  return (size_t)sum_over_ranks((long long)value);
}

size_t cugraph_test_scalar_reduce(const cugraph_resource_handle_t* handle, size_t value) { return cugraph_size_t_allreduce(handle, value); }

int run_mg_test(int (*test)(const cugraph_resource_handle_t*), const char* test_name, const cugraph_resource_handle_t* handle)
{
  time_t t0, t1;
  if (g_rank == 0) {
    printf("RUNNING: %s...", test_name);
    fflush(stdout);
    time(&t0);
  }
  int rc = test(handle);
  rc     = (int)sum_over_ranks(rc);
  if (g_rank == 0) {
    time(&t1);
    printf("done (%f seconds). - %s\n", difftime(t1, t0), rc == 0 ? "passed" : "FAILED");
    fflush(stdout);
  }
  return rc;
}

void* create_mg_raft_handle(int argc, char** argv)
{
  (void)argc; (void)argv;
  char const* e = getenv("CUGRAPH_AMD_TEST_RANKS");
  g_size        = e ? atoi(e) : 2;
  if (g_size < 1 || g_size > 64) g_size = 2;
  char session[64];
  snprintf(session, sizeof(session), "refmg_%ld", (long)getpid());
  fflush(stdout);
  g_rank = 0;
  for (int r = 1; r < g_size; ++r) {
    pid_t const p = fork();
    if (p < 0) { perror("fork"); exit(3); }
    if (p == 0) { g_rank = r; break; }
    g_children[r] = p;
  }
  char const* nd = getenv("CUGRAPH_AMD_TEST_DEVICES");
  if (nd && atoi(nd) > 0) {
    char dev[16];
    snprintf(dev, sizeof(dev), "%d", g_rank % atoi(nd));
    setenv("HIP_VISIBLE_DEVICES", dev, 1);
  }
  setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
  cugraph_error_t* err = NULL;
  if (cugraph_amd_comm_create(session, g_rank, g_size, &g_comm, &err) != CUGRAPH_SUCCESS) {
    printf("rank %d: cugraph_amd_comm_create failed: %s\n", g_rank, err ? cugraph_error_message(err) : "?");
    exit(3);
  }
  return g_comm;
}

void free_mg_raft_handle(void* raft_handle)
{
  cugraph_amd_comm_free((cugraph_amd_comm_t*)raft_handle);
  g_comm = NULL;
  if (g_rank == 0) {  /* (a failing child shows up in rank 0's summed test results; a crashed one makes the collectives time out) */
    for (int r = 1; r < g_size; ++r) {
      int st = 0;
      (void)waitpid(g_children[r], &st, 0);
    }
  }
}

/* host column (rank 0 only) -> owning device array + view (0 on success); the other ranks create empty arrays */
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

int create_mg_test_graph_new(const cugraph_resource_handle_t* handle, cugraph_data_type_id_t vertex_tid, cugraph_data_type_id_t edge_tid, void* h_src,
                             void* h_dst, cugraph_data_type_id_t weight_tid, void* h_wgt, cugraph_data_type_id_t edge_type_tid, void* h_edge_type,
                             cugraph_data_type_id_t edge_id_tid, void* h_edge_id, cugraph_data_type_id_t edge_time_tid, void* h_edge_start_times,
                             void* h_edge_end_times, size_t num_edges, bool_t store_transposed, bool_t renumber, bool_t is_symmetric,
                             bool_t is_multigraph, cugraph_graph_t** graph, cugraph_error_t** ret_error)
{
  (void)edge_tid; (void)renumber;
  cugraph_graph_properties_t properties;
  properties.is_symmetric  = is_symmetric;
  properties.is_multigraph = is_multigraph;
  if (cugraph_resource_handle_get_rank(handle) != 0) num_edges = 0;
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
    cugraph_error_code_t const code = cugraph_graph_create_with_times_mg(
      handle, &properties, NULL, (cugraph_type_erased_device_array_view_t const* const*)&v[0], (cugraph_type_erased_device_array_view_t const* const*)&v[1],
      v[2] ? (cugraph_type_erased_device_array_view_t const* const*)&v[2] : NULL, v[3] ? (cugraph_type_erased_device_array_view_t const* const*)&v[3] : NULL,
      v[4] ? (cugraph_type_erased_device_array_view_t const* const*)&v[4] : NULL, v[5] ? (cugraph_type_erased_device_array_view_t const* const*)&v[5] : NULL,
      v[6] ? (cugraph_type_erased_device_array_view_t const* const*)&v[6] : NULL, store_transposed, 1, FALSE, FALSE, FALSE, FALSE, graph, ret_error);
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

int create_mg_test_graph(const cugraph_resource_handle_t* p_handle, int32_t* h_src, int32_t* h_dst, float* h_wgt, size_t num_edges, bool_t store_transposed,
                         bool_t is_symmetric, cugraph_graph_t** p_graph, cugraph_error_t** ret_error)
{
  return create_mg_test_graph_new(p_handle, INT32, INT32, h_src, h_dst, FLOAT32, h_wgt, INT32, NULL, INT32, NULL, INT32, NULL, NULL, num_edges,
                                  store_transposed, TRUE, is_symmetric, TRUE, p_graph, ret_error);
}

int create_mg_test_graph_double(const cugraph_resource_handle_t* p_handle, int32_t* h_src, int32_t* h_dst, double* h_wgt, size_t num_edges,
                                bool_t store_transposed, bool_t is_symmetric, cugraph_graph_t** p_graph, cugraph_error_t** ret_error)
{
  return create_mg_test_graph_new(p_handle, INT32, INT32, h_src, h_dst, FLOAT64, h_wgt, INT32, NULL, INT32, NULL, INT32, NULL, NULL, num_edges,
                                  store_transposed, TRUE, is_symmetric, TRUE, p_graph, ret_error);
}

int create_mg_test_graph_with_edge_ids(const cugraph_resource_handle_t* p_handle, int32_t* h_src, int32_t* h_dst, int32_t* h_idx, size_t num_edges,
                                       bool_t store_transposed, bool_t is_symmetric, cugraph_graph_t** p_graph, cugraph_error_t** ret_error)
{
  return create_mg_test_graph_new(p_handle, INT32, INT32, h_src, h_dst, FLOAT32, NULL, INT32, NULL, INT32, h_idx, INT32, NULL, NULL, num_edges,
                                  store_transposed, TRUE, is_symmetric, TRUE, p_graph, ret_error);
}

int create_mg_test_graph_with_properties(const cugraph_resource_handle_t* p_handle, int32_t* h_src, int32_t* h_dst, int32_t* h_idx, int32_t* h_type,
                                         float* h_wgt, size_t num_edges, bool_t store_transposed, bool_t is_symmetric, cugraph_graph_t** p_graph,
                                         cugraph_error_t** ret_error)
{
  return create_mg_test_graph_new(p_handle, INT32, INT32, h_src, h_dst, FLOAT32, h_wgt, INT32, h_type, INT32, h_idx, INT32, NULL, NULL, num_edges,
                                  store_transposed, TRUE, is_symmetric, TRUE, p_graph, ret_error);
}
