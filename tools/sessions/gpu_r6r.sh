#!/usr/bin/env bash
# GPU box: BFS at RMAT-24, library of the previous commit against this one (LIBS in gpurun_libs/), timed at the C entry point and at the Python mirror
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
cp cugraph_amd/lib/libcugraph_c.so /tmp/orig.so
: > "$O/${TAG:-r6r}_bfs_ab.txt"
for rep in 1 2 3; do for lib in ${LIBS:-old_bfs new_bfs}; do
  cp "gpurun_libs/$lib.so" cugraph_amd/lib/libcugraph_c.so
  timeout 300 python bench_traversal.py --scale 24 --roots 64 --no-sssp --no-cpu-baseline --no-check 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['bfs']
print('rep $rep lib=$lib  with predecessors: C call mean %.3f ms (min %.3f max %.3f), Python mirror %.3f ms, frac %.4f | distance only: C call %.3f ms frac %.4f' % (b['mean_ms'], b['min_ms'], b['max_ms'], b['python_api_mean_ms'], b['roofline']['frac'], b['distance_only']['mean_ms'], b['distance_only']['roofline_frac']))" | tee -a "$O/${TAG:-r6r}_bfs_ab.txt"
done; done
cp /tmp/orig.so cugraph_amd/lib/libcugraph_c.so
