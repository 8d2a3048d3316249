#!/usr/bin/env bash
# GPU box: SSSP tests + bench_traversal.py (RMAT-24, integer weights) for several settings of the light/heavy bucket path
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q -k "${PYTEST_K:-sssp}" 2>&1 | tail -4
for envs in ${ENVS:-CUGRAPH_AMD_SSSP_LH=0 CUGRAPH_AMD_SSSP_LH=1}; do
  echo "== $envs"
  env $(echo $envs | tr ',' ' ') timeout 300 python bench_traversal.py --scale 24 --roots ${ROOTS:-16} --weights ${WEIGHTS:-int} --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); s=j['sssp']; print('sssp mean_ms', s['mean_ms'], 'min', s['min_ms'], 'max', s['max_ms'], 'steps', s.get('mean_steps'), 'relaxations/E', s.get('mean_relaxations_per_edge'))"
done
if [ -n "${PROF_ENV:-}" ]; then
  P="$O/prof_sssp"; rm -rf "$P"; mkdir -p "$P"
  ( cd /tmp && export TMPDIR=/tmp && env $(echo $PROF_ENV | tr ',' ' ') timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$P/stats" -o run -- python $R/bench_traversal.py --scale 24 --roots 4 --weights int --no-cpu-baseline > "$P/stats.log" 2>&1 )
  python tools/rocpd_summary.py "$P" > "$P/summary.txt" 2>&1; find "$P" -name "*.db" -delete
  head -14 "$P/summary.txt" | cut -c1-150
fi
