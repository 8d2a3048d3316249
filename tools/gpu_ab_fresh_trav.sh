#!/usr/bin/env bash
# GPU box: library variants (LIBS) on bench_traversal.py (RMAT-24, ROOTS roots, WEIGHTS) in fresh processes, alternating (REPS rounds)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
cp cugraph_amd/lib/libcugraph_c.so /tmp/orig.so
out="$O/${TAG:-ab}_ab_trav.txt"; : > "$out"
for rep in $(seq 1 ${REPS:-3}); do for lib in ${LIBS:-cur}; do
  cp "gpurun_libs/$lib.so" cugraph_amd/lib/libcugraph_c.so
  echo -n "rep $rep lib $lib: " | tee -a "$out"
  timeout 300 python bench_traversal.py --scale ${SCALE:-24} --weights ${WEIGHTS:-int} --roots ${ROOTS:-32} --no-cpu-baseline ${EXTRA:-} 2>/dev/null | python tools/trav_line.py | tee -a "$out"
done; done
cp /tmp/orig.so cugraph_amd/lib/libcugraph_c.so
