#!/usr/bin/env bash
# Builds an importable `pylibcugraph` package out of the REFERENCE's unchanged Cython modules (cythonized in place from
# $CUGRAPH_REFERENCE_DIR/python/pylibcugraph, nothing copied) linked to cugraph_amd/lib/libcugraph_c.so, plus the stand-ins of
# tests/pylibcugraph_run/standins (a package shell, exceptions, one api_tools function, and a torch-backed sliver of cupy).
# Output: tests/pylibcugraph_run/_pkg/ (git-ignored; travels to the GPU box with the repo snapshot).
set -eu
R="$(cd "$(dirname "$0")/../.." && pwd)"
REF="${CUGRAPH_REFERENCE_DIR:-/root/reference}/python/pylibcugraph"
O="$R/tests/pylibcugraph_run/_pkg"
[ -d "$REF/pylibcugraph" ] || { echo "no reference tree at $REF: nothing to build"; exit 0; }
rm -rf "$O"; mkdir -p "$O/build" "$O/inc/dlpack"
cp -r "$R/tests/pylibcugraph_run/standins/." "$O/"
DL="$(python -c 'import torch, os; print(os.path.join(os.path.dirname(torch.__file__), "include", "ATen", "dlpack.h"))')"
[ -f "$DL" ] && cp "$DL" "$O/inc/dlpack/dlpack.h"
PYINC="$(python -c 'import sysconfig; print(sysconfig.get_paths()["include"])')"
SUF="$(python -c 'import sysconfig; print(sysconfig.get_config_var("EXT_SUFFIX"))')"
for m in graphs utils resource_handle graph_properties pagerank personalized_pagerank bfs sssp has_vertex louvain degrees decompress_to_edgelist random generate_rmat_edgelist; do
  python -m cython -3 --cplus -I "$REF" -o "$O/build/$m.cpp" "$REF/pylibcugraph/$m.pyx"
  g++ -O1 -w -fPIC -shared -std=c++17 -I"$R/include" -I"$O/inc" -I"$PYINC" "$O/build/$m.cpp" -o "$O/pylibcugraph/$m$SUF" \
      -L"$R/cugraph_amd/lib" -lcugraph_c -Wl,-rpath,'$ORIGIN/../../../../cugraph_amd/lib' &
done
wait
rm -rf "$O/build"
ls "$O/pylibcugraph" | wc -l
