/* Plain-C driver of the multi-GPU entry points (gcc -std=c99, no HIP headers): N processes forked from one binary, each a rank on
 * the library's communicator (include/cugraph_amd/extensions.h: cugraph_amd_comm_create -> cugraph_create_resource_handle(comm)),
 * each holding a SLICE of the 6-vertex graph of the reference's C tests; cugraph_graph_create_mg, then cugraph_pagerank, cugraph_bfs and
 * cugraph_sssp -- collective calls, every rank gets the vertices it owns back -- and the union of the ranks' answers must equal the
 * single-GPU goldens (cpp/tests/c_api/pagerank_test.c:385-480, bfs_test.c test_bfs, sssp_test.c tail; the reference's own MG tests,
 * cpp/tests/c_api/mg_pagerank_test.c / mg_bfs_test.c / mg_sssp_test.c, check the same graph against the same values).
 * usage: mg_two_ranks [n_ranks]   (all ranks share HIP device 0 unless there are enough devices).  Exit code 0 = every rank passed. */
#define _POSIX_C_SOURCE 200809L
#include <cugraph_amd/extensions.h>
#include <cugraph_c/array.h>
#include <cugraph_c/centrality_algorithms.h>
#include <cugraph_c/error.h>
#include <cugraph_c/graph.h>
#include <cugraph_c/resource_handle.h>
#include <cugraph_c/traversal_algorithms.h>

#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#define CHECK(cond, msg)                                                                                       \
  do {                                                                                                         \
    if (!(cond)) { fprintf(stderr, "rank %d FAILED %s:%d %s\n", rank, __FILE__, __LINE__, msg); return 1; }      \
  } while (0)
#define OK(call)                                                                                               \
  do {                                                                                                         \
    cugraph_error_code_t c_ = (call);                                                                          \
    if (c_ != CUGRAPH_SUCCESS) { fprintf(stderr, "rank %d FAILED %s:%d code %d: %s\n", rank, __FILE__, __LINE__, (int)c_, err ? cugraph_error_message(err) : "?"); return 1; } \
  } while (0)

static int rank = 0;

static int upload(const cugraph_resource_handle_t* h, const void* host, size_t n, cugraph_data_type_id_t t, cugraph_type_erased_device_array_t** arr,
                  cugraph_type_erased_device_array_view_t** view)
{
  cugraph_error_t* err = NULL;
  OK(cugraph_type_erased_device_array_create(h, n, t, arr, &err));
  *view = cugraph_type_erased_device_array_view(*arr);
  if (n) OK(cugraph_type_erased_device_array_view_copy_from_host(h, *view, (const byte_t*)host, &err));
  return 0;
}

static int download(const cugraph_resource_handle_t* h, cugraph_type_erased_device_array_view_t* v, void* host)
{
  cugraph_error_t* err = NULL;
  if (cugraph_type_erased_device_array_view_size(v)) OK(cugraph_type_erased_device_array_view_copy_to_host(h, (byte_t*)host, v, &err));
  cugraph_type_erased_device_array_view_free(v);
  return 0;
}

static int near(double a, double b, double tol) { return fabs(a - b) <= tol; }

static int run_rank(const char* session, int size)
{
  cugraph_error_t* err = NULL;
  cugraph_amd_comm_t* comm = NULL;
  OK(cugraph_amd_comm_create(session, rank, size, &comm, &err));
  cugraph_resource_handle_t* h = cugraph_create_resource_handle((void*)comm);  /* the place of the raft::handle_t* in the reference */
  CHECK(h != NULL, "resource handle on the communicator");
  CHECK(cugraph_resource_handle_get_rank(h) == rank && cugraph_resource_handle_get_comm_size(h) == size, "rank / size of the handle");

  /* the graph of pagerank_test.c / bfs_test.c / sssp_test.c; rank r holds the edges e with e % size == r */
  int32_t src[] = {0, 1, 1, 2, 2, 2, 3, 4}, dst[] = {1, 3, 4, 0, 1, 3, 5, 5};
  float w[]     = {0.1f, 2.1f, 1.1f, 5.1f, 3.1f, 4.1f, 7.2f, 3.2f};
  int32_t ms[8], md[8];
  float mw[8];
  size_t m = 0;
  for (int e = 0; e < 8; ++e)
    if (e % size == rank) { ms[m] = src[e]; md[m] = dst[e]; mw[m] = w[e]; ++m; }
  cugraph_type_erased_device_array_t *as, *ad, *aw;
  cugraph_type_erased_device_array_view_t *vs, *vd, *vw;
  if (upload(h, ms, m, INT32, &as, &vs) || upload(h, md, m, INT32, &ad, &vd) || upload(h, mw, m, FLOAT32, &aw, &vw)) return 1;
  cugraph_type_erased_device_array_view_t const* srcs[1] = {vs};
  cugraph_type_erased_device_array_view_t const* dsts[1] = {vd};
  cugraph_type_erased_device_array_view_t const* wgts[1] = {vw};
  cugraph_graph_properties_t props = {FALSE, FALSE};
  cugraph_graph_t* g = NULL;
  OK(cugraph_graph_create_mg(h, &props, NULL, srcs, dsts, wgts, NULL, NULL, TRUE, 1, FALSE, FALSE, FALSE, FALSE, &g, &err));

  /* PageRank: alpha 0.95, epsilon 1e-4, 20 iterations (pagerank_test.c:385-400), tolerance of the reference's test: 1e-3 */
  double const pr_gold[6] = {0.0915528, 0.168382, 0.0656831, 0.191468, 0.120677, 0.362237};
  cugraph_centrality_result_t* cres = NULL;
  OK(cugraph_pagerank(h, g, NULL, NULL, NULL, NULL, 0.95, 0.0001, 20, FALSE, &cres, &err));
  size_t n_own = cugraph_type_erased_device_array_view_size(cugraph_centrality_result_get_vertices(cres));
  int32_t v[6];
  float x[6];
  CHECK(n_own <= 6, "owned vertices");
  if (download(h, cugraph_centrality_result_get_vertices(cres), v) || download(h, cugraph_centrality_result_get_values(cres), x)) return 1;
  for (size_t i = 0; i < n_own; ++i) CHECK(v[i] >= 0 && v[i] < 6 && near(x[i], pr_gold[v[i]], 1e-3), "pagerank value");
  cugraph_centrality_result_free(cres);
  size_t pr_rows = n_own;

  /* BFS from 0 (rank 0 names the source), depth limit 10: bfs_test.c test_bfs */
  int32_t const bfs_dist[6] = {0, 1, 2147483647, 2, 2, 3}, bfs_pred[6] = {-1, 0, -1, 1, 1, 3};
  int32_t seed0[1] = {0};
  cugraph_type_erased_device_array_t* aseed;
  cugraph_type_erased_device_array_view_t* vseed;
  if (upload(h, seed0, rank == 0 ? 1 : 0, INT32, &aseed, &vseed)) return 1;
  cugraph_paths_result_t* pres = NULL;
  OK(cugraph_bfs(h, g, vseed, FALSE, 10, TRUE, FALSE, &pres, &err));
  int32_t dd[6], pp[6];
  n_own = cugraph_type_erased_device_array_view_size(cugraph_paths_result_get_vertices(pres));
  CHECK(n_own <= 6, "owned vertices (bfs)");
  if (download(h, cugraph_paths_result_get_vertices(pres), v) || download(h, cugraph_paths_result_get_distances(pres), dd) || download(h, cugraph_paths_result_get_predecessors(pres), pp)) return 1;
  for (size_t i = 0; i < n_own; ++i) CHECK(dd[i] == bfs_dist[v[i]] && pp[i] == bfs_pred[v[i]], "bfs distance / predecessor");
  cugraph_paths_result_free(pres);
  size_t bfs_rows = n_own;

  /* SSSP from 0: sssp_test.c */
  float const sssp_dist[6] = {0.0f, 0.1f, FLT_MAX, 2.2f, 1.2f, 4.4f};
  int32_t const sssp_pred[6] = {-1, 0, -1, 1, 1, 4};
  float fd[6];
  OK(cugraph_sssp(h, g, 0, (double)FLT_MAX, TRUE, FALSE, &pres, &err));
  n_own = cugraph_type_erased_device_array_view_size(cugraph_paths_result_get_vertices(pres));
  if (download(h, cugraph_paths_result_get_vertices(pres), v) || download(h, cugraph_paths_result_get_distances(pres), fd) || download(h, cugraph_paths_result_get_predecessors(pres), pp)) return 1;
  for (size_t i = 0; i < n_own; ++i) CHECK(near(fd[i], sssp_dist[v[i]], 1e-5 * fmax(1.0, sssp_dist[v[i]] == FLT_MAX ? 1.0 : sssp_dist[v[i]])) && pp[i] == sssp_pred[v[i]], "sssp distance / predecessor");
  cugraph_paths_result_free(pres);

  /* every vertex came back from exactly one rank: the owned-row counts add up to 6 for each algorithm */
  double mine[3] = {(double)pr_rows, (double)bfs_rows, (double)n_own}, all[3 * 64];
  OK(cugraph_amd_comm_host_allgather(comm, mine, sizeof(mine), all, &err));
  double t0 = 0, t1 = 0, t2 = 0;
  for (int r = 0; r < size; ++r) { t0 += all[3 * r]; t1 += all[3 * r + 1]; t2 += all[3 * r + 2]; }
  CHECK(t0 == 6 && t1 == 6 && t2 == 6, "the ranks' owned vertices must partition the vertex set");

  cugraph_type_erased_device_array_view_free(vseed); cugraph_type_erased_device_array_free(aseed);
  cugraph_graph_free(g);
  cugraph_type_erased_device_array_view_free(vs); cugraph_type_erased_device_array_view_free(vd); cugraph_type_erased_device_array_view_free(vw);
  cugraph_type_erased_device_array_free(as); cugraph_type_erased_device_array_free(ad); cugraph_type_erased_device_array_free(aw);
  cugraph_free_resource_handle(h);
  cugraph_amd_comm_free(comm);
  return 0;
}

int main(int argc, char** argv)
{
  int size = argc > 1 ? atoi(argv[1]) : 2;
  if (size < 1 || size > 8) size = 2;
  char session[64];
  snprintf(session, sizeof(session), "ctest_%ld", (long)getpid());
  pid_t pids[8];
  for (int r = 0; r < size; ++r) {  /* fork BEFORE anything touches the GPU: one process per rank */
    pids[r] = fork();
    if (pids[r] == 0) {
      rank = r;
      _exit(run_rank(session, size));
    }
  }
  int bad = 0;
  for (int r = 0; r < size; ++r) {
    int st = 0;
    waitpid(pids[r], &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { fprintf(stderr, "rank %d exited with status %d\n", r, st); bad = 1; }
  }
  if (!bad) printf("c_api multi-rank: ok (%d ranks)\n", size);
  return bad;
}
