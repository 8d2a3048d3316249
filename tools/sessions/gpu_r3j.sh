#!/usr/bin/env bash
# GPU box, round 3 session J: device-side small BFS levels (parity + time, A/B by env), new phase-1 schedule defaults, Louvain-26 with the pool cap restored
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee "$O/r3j_pytest.log"
for sm in 1 0; do
  CUGRAPH_AMD_BFS_SMALL=$sm timeout 300 python bench_traversal.py --scale 24 --roots 32 --no-sssp --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['bfs']; print('small=$sm', 'bfs mean', b['mean_ms'], 'min', b['min_ms'], 'max', b['max_ms'], 'GTEPS', b['harmonic_mean_mteps']/1e3, 'levels', b['mean_levels'], 'check', b['check']['ok'])"
  CUGRAPH_AMD_BFS_SMALL=$sm timeout 300 python bench_traversal.py --scale 24 --roots 32 --no-sssp --no-cpu-baseline --symmetric 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['bfs']; print('small=$sm sym', 'bfs mean', b['mean_ms'], 'min', b['min_ms'], 'max', b['max_ms'], 'GTEPS', b['harmonic_mean_mteps']/1e3, 'check', b['check']['ok'])"
  CUGRAPH_AMD_BFS_SMALL=$sm timeout 300 python bench_traversal.py --scale 24 --roots 32 --no-sssp --no-cpu-baseline --predecessors 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['bfs']; print('small=$sm pred', 'bfs mean', b['mean_ms'], 'min', b['min_ms'], 'max', b['max_ms'], 'check', b['check']['ok'])"
done
CUGRAPH_AMD_BFS_TRACE=1 timeout 300 python bench_traversal.py --scale 24 --roots 1 --no-sssp --no-cpu-baseline --no-check 2>&1 | grep "\[bfs\]" | tail -12
for sc in 22 24; do
  timeout 300 python bench.py --scale $sc --no-extras --cpu-scale 20 2>/dev/null | tee "$O/r3j_bench_s$sc.json" | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('s$sc ms/step', d['ms_per_step'], 'frac', r['frac'], 'p1', r['avg_phase1_ms'], 'p2', r['avg_phase2_ms'], 'check', d['check']['ok'], 'build', d['graph_build_s'], d['plan_build_s'])"
done
CUGRAPH_AMD_LOUVAIN_TRACE=1 timeout 600 python bench_louvain.py --scale 26 --cpu-scale 0 --repeats 2 --out "$O/r3j_louvain_s26.json" 2>&1 | grep -E "\[louvain\] [0-9]|\"value\"" | tail -4 | cut -c1-200
