#!/usr/bin/env bash
# GPU box: Louvain chunk kernel, threads per workgroup x edges per thread (LIBS = variants in gpurun_libs/, "base" = the tree's library), alternating, RMAT-22
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${TAG:-r6x}
cp cugraph_amd/lib/libcugraph_c.so /tmp/orig.so; cp /tmp/orig.so gpurun_libs/base.so
: > "$O/${TAG}_louvain_ab.txt"
for rep in 1 2; do for lib in ${LIBS:-base}; do
  cp "gpurun_libs/$lib.so" cugraph_amd/lib/libcugraph_c.so
  echo "== rep $rep lib=$lib" >> "$O/${TAG}_louvain_ab.txt"
  CUGRAPH_AMD_LOUVAIN_TRACE=1 timeout 600 python bench_louvain.py --scale 22 --cpu-scale 0 --repeats 3 2>&1 | grep -E "louvain\]|^\{" | tail -7 | cut -c1-330 >> "$O/${TAG}_louvain_ab.txt"
done; done
cp /tmp/orig.so cugraph_amd/lib/libcugraph_c.so
grep -E "==|levels|^\{" "$O/${TAG}_louvain_ab.txt" | cut -c1-120
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "louvain" 2>&1 | tail -3
