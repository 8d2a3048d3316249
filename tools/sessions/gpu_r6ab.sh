#!/usr/bin/env bash
# round 6: A/B of library variants in gpurun_libs/ (LIBS) on SSSP (RMAT-24, WEIGHTS) with the result check on; then the SSSP parity tests on the tree's library
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${TAG:-r6ab}
cp cugraph_amd/lib/libcugraph_c.so /tmp/orig.so
{
for rep in 1 2; do for lib in ${LIBS:-base sweep3}; do for wt in ${WEIGHTS:-int unit}; do
  cp "gpurun_libs/$lib.so" cugraph_amd/lib/libcugraph_c.so
  echo "== rep $rep lib=$lib weights=$wt"
  timeout 300 python bench_traversal.py --scale 24 --roots ${ROOTS:-16} --weights $wt --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); d=j['sssp']; print('sssp mean_ms', d['mean_ms'], 'min', d['min_ms'], 'max', d['max_ms'], 'check', (d.get('check') or {}).get('ok'), 'relax/edge', d.get('relaxations_per_edge'), 'bfs', j['bfs']['mean_ms'])"
done; done; done
} 2>&1 | tee "$O/${TAG}_sssp_ab.txt"
cp /tmp/orig.so cugraph_amd/lib/libcugraph_c.so
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sssp" 2>&1 | tail -5 | tee "$O/${TAG}_pytest_sssp.log"
