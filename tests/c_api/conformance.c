/* Plain-C caller of the drop-in C ABI (gcc -std=c99, no HIP headers): the 6-vertex graph the reference's C tests use,
 * with their expected values (cpp/tests/c_api/pagerank_test.c:385-480, bfs_test.c test_bfs, sssp_test.c tail; values in
 * tests/golden/golden.json).  Host data goes in and out through cugraph_type_erased_device_array_view_copy_from_host /
 * _copy_to_host, exactly as those tests do.  Exit code 0 = all checks passed. */
#include <cugraph_c/array.h>
#include <cugraph_c/centrality_algorithms.h>
#include <cugraph_c/error.h>
#include <cugraph_c/graph.h>
#include <cugraph_c/graph_functions.h>
#include <cugraph_c/resource_handle.h>
#include <cugraph_c/traversal_algorithms.h>

#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(cond, msg)                                                     \
  do {                                                                       \
    if (!(cond)) { fprintf(stderr, "FAILED %s:%d %s\n", __FILE__, __LINE__, msg); return 1; } \
  } while (0)
#define OK(call)                                                                                   \
  do {                                                                                             \
    cugraph_error_t* e_ = NULL;                                                                    \
    cugraph_error_code_t c_ = (call);                                                              \
    if (c_ != CUGRAPH_SUCCESS) { fprintf(stderr, "FAILED %s:%d code %d: %s\n", __FILE__, __LINE__, (int)c_, err ? cugraph_error_message(err) : "?"); return 1; } \
    (void)e_;                                                                                      \
  } while (0)

static int upload(const cugraph_resource_handle_t* h, const void* host, size_t n, cugraph_data_type_id_t t,
                  cugraph_type_erased_device_array_t** arr, cugraph_type_erased_device_array_view_t** view)
{
  cugraph_error_t* err = NULL;
  OK(cugraph_type_erased_device_array_create(h, n, t, arr, &err));
  *view = cugraph_type_erased_device_array_view(*arr);
  OK(cugraph_type_erased_device_array_view_copy_from_host(h, *view, (const byte_t*)host, &err));
  return 0;
}

static int make_graph(const cugraph_resource_handle_t* h, bool_t transposed, bool_t renumber, cugraph_graph_t** g)
{
  int32_t src[] = {0, 1, 1, 2, 2, 2, 3, 4}, dst[] = {1, 3, 4, 0, 1, 3, 5, 5};
  float w[]     = {0.1f, 2.1f, 1.1f, 5.1f, 3.1f, 4.1f, 7.2f, 3.2f};
  cugraph_type_erased_device_array_t *as, *ad, *aw;
  cugraph_type_erased_device_array_view_t *vs, *vd, *vw;
  cugraph_error_t* err = NULL;
  if (upload(h, src, 8, INT32, &as, &vs) || upload(h, dst, 8, INT32, &ad, &vd) || upload(h, w, 8, FLOAT32, &aw, &vw)) return 1;
  cugraph_graph_properties_t props = {FALSE, FALSE};
  OK(cugraph_graph_create_sg(h, &props, NULL, vs, vd, vw, NULL, NULL, transposed, renumber, FALSE, FALSE, FALSE, FALSE, g, &err));
  cugraph_type_erased_device_array_view_free(vs); cugraph_type_erased_device_array_view_free(vd); cugraph_type_erased_device_array_view_free(vw);
  cugraph_type_erased_device_array_free(as); cugraph_type_erased_device_array_free(ad); cugraph_type_erased_device_array_free(aw);
  return 0;
}

static int near(double a, double b, double rel) { return fabs(a - b) <= rel * fmax(fabs(a), fabs(b)); }

int main(void)
{
  cugraph_error_t* err         = NULL;
  cugraph_resource_handle_t* h = cugraph_create_resource_handle(NULL);
  CHECK(h != NULL, "resource handle");
  for (int variant = 0; variant < 4; ++variant) {
    bool_t transposed = (variant & 1) ? TRUE : FALSE, renumber = (variant & 2) ? TRUE : FALSE;
    cugraph_graph_t* g = NULL;
    if (make_graph(h, transposed, renumber, &g)) return 1;

    /* PageRank: alpha 0.95, epsilon 1e-4, 20 iterations (pagerank_test.c test_pagerank) */
    {
      double expect[] = {0.0915528, 0.168382, 0.0656831, 0.191468, 0.120677, 0.362237};
      cugraph_centrality_result_t* r = NULL;
      OK(cugraph_pagerank(h, g, NULL, NULL, NULL, NULL, 0.95, 0.0001, 20, FALSE, &r, &err));
      int32_t v[6]; float pr[6];
      cugraph_type_erased_device_array_view_t* vv = cugraph_centrality_result_get_vertices(r);
      cugraph_type_erased_device_array_view_t* pv = cugraph_centrality_result_get_values(r);
      CHECK(cugraph_type_erased_device_array_view_size(vv) == 6 && cugraph_type_erased_device_array_view_type(pv) == FLOAT32, "result shape");
      OK(cugraph_type_erased_device_array_view_copy_to_host(h, (byte_t*)v, vv, &err));
      OK(cugraph_type_erased_device_array_view_copy_to_host(h, (byte_t*)pr, pv, &err));
      for (int i = 0; i < 6; ++i) CHECK(v[i] >= 0 && v[i] < 6 && near(pr[i], expect[v[i]], 1e-3), "pagerank value");
      CHECK(cugraph_centrality_result_converged(r) == TRUE, "converged");
      cugraph_type_erased_device_array_view_free(vv); cugraph_type_erased_device_array_view_free(pv);
      cugraph_centrality_result_free(r);
      /* 2 iterations do not converge: cugraph_pagerank reports an error but still hands the result over */
      cugraph_error_code_t c = cugraph_pagerank(h, g, NULL, NULL, NULL, NULL, 0.95, 0.0001, 2, FALSE, &r, &err);
      CHECK(c == CUGRAPH_UNKNOWN_ERROR && r != NULL && err != NULL, "non-convergence must be reported");
      cugraph_error_free(err); err = NULL;
      cugraph_centrality_result_free(r);
    }
    /* BFS from 0, depth limit 10 (bfs_test.c test_bfs) */
    {
      int32_t seed = 0, ed[] = {0, 1, 2147483647, 2, 2, 3}, ep[] = {-1, 0, -1, 1, 1, 3};
      cugraph_type_erased_device_array_t* as; cugraph_type_erased_device_array_view_t* vs;
      if (upload(h, &seed, 1, INT32, &as, &vs)) return 1;
      cugraph_paths_result_t* r = NULL;
      OK(cugraph_bfs(h, g, vs, FALSE, 10, TRUE, FALSE, &r, &err));
      int32_t v[6], d[6], p[6];
      cugraph_type_erased_device_array_view_t *vv = cugraph_paths_result_get_vertices(r), *dv = cugraph_paths_result_get_distances(r),
                                              *pv = cugraph_paths_result_get_predecessors(r);
      OK(cugraph_type_erased_device_array_view_copy_to_host(h, (byte_t*)v, vv, &err));
      OK(cugraph_type_erased_device_array_view_copy_to_host(h, (byte_t*)d, dv, &err));
      OK(cugraph_type_erased_device_array_view_copy_to_host(h, (byte_t*)p, pv, &err));
      for (int i = 0; i < 6; ++i) CHECK(d[i] == ed[v[i]] && p[i] == ep[v[i]], "bfs distance / predecessor");
      cugraph_type_erased_device_array_view_free(vv); cugraph_type_erased_device_array_view_free(dv); cugraph_type_erased_device_array_view_free(pv);
      /* extract_paths_test.c test_bfs_with_extract_paths: destination 5 -> max path length 4, path 0 1 3 5 */
      {
        int32_t dest = 5, path[4], ex[] = {0, 1, 3, 5};
        cugraph_type_erased_device_array_t* ad; cugraph_type_erased_device_array_view_t* vdst;
        if (upload(h, &dest, 1, INT32, &ad, &vdst)) return 1;
        cugraph_extract_paths_result_t* er = NULL;
        OK(cugraph_extract_paths(h, g, vs, r, vdst, &er, &err));
        CHECK(cugraph_extract_paths_result_get_max_path_length(er) == 4, "max path length");
        cugraph_type_erased_device_array_view_t* pth = cugraph_extract_paths_result_get_paths(er);
        CHECK(cugraph_type_erased_device_array_view_size(pth) == 4, "paths size");
        OK(cugraph_type_erased_device_array_view_copy_to_host(h, (byte_t*)path, pth, &err));
        for (int i = 0; i < 4; ++i) CHECK(path[i] == ex[i], "extracted path");
        cugraph_type_erased_device_array_view_free(pth);
        cugraph_extract_paths_result_free(er);
        cugraph_type_erased_device_array_view_free(vdst); cugraph_type_erased_device_array_free(ad);
      }
      cugraph_paths_result_free(r);
      cugraph_type_erased_device_array_view_free(vs); cugraph_type_erased_device_array_free(as);
    }
    /* degrees (degrees_test.c test_degrees) */
    {
      int32_t ein[] = {1, 2, 0, 2, 1, 2}, eout[] = {1, 2, 3, 1, 1, 0}, v[6], di[6], dout[6];
      cugraph_degrees_result_t* dr = NULL;
      OK(cugraph_degrees(h, g, NULL, FALSE, &dr, &err));
      cugraph_type_erased_device_array_view_t *vv = cugraph_degrees_result_get_vertices(dr), *iv = cugraph_degrees_result_get_in_degrees(dr),
                                              *ov = cugraph_degrees_result_get_out_degrees(dr);
      CHECK(vv != NULL && iv != NULL && ov != NULL && cugraph_type_erased_device_array_view_size(vv) == 6, "degrees result shape");
      OK(cugraph_type_erased_device_array_view_copy_to_host(h, (byte_t*)v, vv, &err));
      OK(cugraph_type_erased_device_array_view_copy_to_host(h, (byte_t*)di, iv, &err));
      OK(cugraph_type_erased_device_array_view_copy_to_host(h, (byte_t*)dout, ov, &err));
      for (int i = 0; i < 6; ++i) CHECK(di[i] == ein[v[i]] && dout[i] == eout[v[i]], "in / out degree");
      cugraph_type_erased_device_array_view_free(vv); cugraph_type_erased_device_array_view_free(iv); cugraph_type_erased_device_array_view_free(ov);
      cugraph_degrees_result_free(dr);
    }
    /* SSSP from 0 (sssp_test.c test_sssp) */
    {
      float ed[] = {0.0f, 0.1f, FLT_MAX, 2.2f, 1.2f, 4.4f};
      int32_t ep[] = {-1, 0, -1, 1, 1, 4};
      cugraph_paths_result_t* r = NULL;
      OK(cugraph_sssp(h, g, 0, DBL_MAX, TRUE, FALSE, &r, &err));
      int32_t v[6], p[6]; float d[6];
      cugraph_type_erased_device_array_view_t *vv = cugraph_paths_result_get_vertices(r), *dv = cugraph_paths_result_get_distances(r),
                                              *pv = cugraph_paths_result_get_predecessors(r);
      OK(cugraph_type_erased_device_array_view_copy_to_host(h, (byte_t*)v, vv, &err));
      OK(cugraph_type_erased_device_array_view_copy_to_host(h, (byte_t*)d, dv, &err));
      OK(cugraph_type_erased_device_array_view_copy_to_host(h, (byte_t*)p, pv, &err));
      for (int i = 0; i < 6; ++i) CHECK(near(d[i], ed[v[i]], 1e-6) && p[i] == ep[v[i]], "sssp distance / predecessor");
      cugraph_type_erased_device_array_view_free(vv); cugraph_type_erased_device_array_view_free(dv); cugraph_type_erased_device_array_view_free(pv);
      cugraph_paths_result_free(r);
    }
    cugraph_graph_free(g);
  }
  cugraph_free_resource_handle(h);
  printf("c_api conformance: ok\n");
  return 0;
}
