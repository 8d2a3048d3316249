#!/usr/bin/env bash
# Compiles the reference's plain-C API tests IN PLACE (sources stay under $CUGRAPH_REFERENCE_DIR; nothing is copied) against
# include/ and links them to cugraph_amd/lib/libcugraph_c.so.  Outputs: tests/c_api/_ref_bin/<name> (git-ignored; they travel
# to the GPU box with the repo snapshot, where tests/test_reference_c_tests.py runs them).
set -eu
R="$(cd "$(dirname "$0")/../.." && pwd)"
REF="${CUGRAPH_REFERENCE_DIR:-/root/reference}"
T="$REF/cpp/tests/c_api"
O="$R/tests/c_api/_ref_bin"
[ -d "$T" ] || { echo "no reference tree at $REF: nothing to build"; exit 0; }
mkdir -p "$O"
CFLAGS="-std=gnu99 -O1 -w -I$R/include -I$R/tests/c_api/ref_shim -I$T"
LDFLAGS="-L$R/cugraph_amd/lib -lcugraph_c -lm -Wl,-rpath,\$ORIGIN/../../../cugraph_amd/lib -Wl,--no-undefined"
for name in pagerank_test bfs_test sssp_test louvain_test degrees_test extract_paths_test; do
  gcc $CFLAGS "$T/$name.c" "$R/tests/c_api/ref_test_shim.c" -o "$O/$name" $LDFLAGS
done
gcc $CFLAGS -Dmain=reference_main "$T/create_graph_test.c" -c -o "$O/create_graph_test.o"
gcc $CFLAGS -DCGA_CREATE_GRAPH_MAIN "$R/tests/c_api/ref_test_shim.c" "$O/create_graph_test.o" -o "$O/create_graph_test" $LDFLAGS
rm -f "$O/create_graph_test.o"
# the reference's multi-GPU tests: the same sources a maintainer runs under mpirun; here the helper library forks the ranks
# (tests/c_api/ref_mg_test_shim.c) on this library's own communicator
for name in mg_pagerank_test mg_bfs_test mg_sssp_test mg_louvain_test mg_create_graph_test mg_degrees_test mg_generate_rmat_test; do
  gcc $CFLAGS "$T/$name.c" "$R/tests/c_api/ref_mg_test_shim.c" -o "$O/$name" $LDFLAGS
done
ls "$O"
