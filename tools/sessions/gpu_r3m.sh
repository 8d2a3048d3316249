#!/usr/bin/env bash
# GPU box, round 3 session M: SSSP with distance-ordered sub-queues (parity + time against the single near queue, K sweep)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sssp or goldens or conformance" 2>&1 | tail -4
timeout 300 python -m pytest tests/test_c_conformance.py tests/test_reference_c_tests.py -m gpu -q -x 2>&1 | tail -3
run() { # label, env...
  label="$1"; shift
  env "$@" timeout 300 python bench_traversal.py --scale 24 --roots 16 --weights "$W" --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['sssp']; print('$label W=$W', 'sssp mean', s['mean_ms'], 'min', s['min_ms'], 'max', s['max_ms'], 'steps', s['mean_steps'], 'relax/edge', s['mean_relaxations_per_edge'], 'check', s['check']['ok'], s.get('unit_weight_distances_equal_bfs'))"
}
for W in int unit; do
  run nearfar CUGRAPH_AMD_SSSP_MODE=nearfar
  run multi8 CUGRAPH_AMD_SSSP_SUBQ=8
  run multi4 CUGRAPH_AMD_SSSP_SUBQ=4
  run multi2 CUGRAPH_AMD_SSSP_SUBQ=2
  run multi1 CUGRAPH_AMD_SSSP_SUBQ=1
done
W=int
run multi8_d2 CUGRAPH_AMD_SSSP_SUBQ=8 CUGRAPH_AMD_SSSP_DELTA_SCALE=2
run multi8_d05 CUGRAPH_AMD_SSSP_SUBQ=8 CUGRAPH_AMD_SSSP_DELTA_SCALE=0.5
run multi4_d05 CUGRAPH_AMD_SSSP_SUBQ=4 CUGRAPH_AMD_SSSP_DELTA_SCALE=0.5
