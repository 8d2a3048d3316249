#!/usr/bin/env bash
# GPU box: library variants (LIBS) on bench_louvain.py at SCALE in fresh processes, alternating (REPS rounds)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
cp cugraph_amd/lib/libcugraph_c.so /tmp/orig.so
out="$O/${TAG:-ab}_ab_louvain.txt"; : > "$out"
for rep in $(seq 1 ${REPS:-4}); do for lib in ${LIBS:-cur}; do
  cp "gpurun_libs/$lib.so" cugraph_amd/lib/libcugraph_c.so
  timeout 300 python bench_louvain.py --scale ${SCALE:-22} --cpu-scale 0 --repeats 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('rep $rep lib $lib seconds', d['value'], 'all', d['seconds_all'], 'ok', d['check']['ok'])" | tee -a "$out"
done; done
cp /tmp/orig.so cugraph_amd/lib/libcugraph_c.so
