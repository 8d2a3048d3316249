#!/usr/bin/env bash
# round 4, session l: BFS with one-atomic-per-workgroup bitmap -> queue and chained tail levels: parity tests, then the Graph500-protocol line
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_mg_traversal.py tests/test_mg_capi.py -m gpu -x -q -k "bfs" 2>&1 | tail -6 | tee "$O/r4l_bfs_tests.log"
for chain in 1; do
  CUGRAPH_AMD_BFS_CHAIN=$chain timeout 600 python bench_traversal.py --scale 24 --weights int --roots 32 --no-sssp --no-cpu-baseline --out "$O/r4l_bfs_s24_chain$chain.json" > /dev/null 2>&1
  python - <<PY
import json
d=json.load(open("$O/r4l_bfs_s24_chain$chain.json"))
b=d["bfs"]; print("chain $chain", "mean_ms", b.get("mean_ms"), "min/max", b.get("min_ms"), b.get("max_ms"), "gteps", b.get("harmonic_mean_mteps"), "frac", b["roofline"]["frac"], "check", (b.get("check") or {}).get("ok"))
PY
done
CUGRAPH_AMD_BFS_TRACE=1 timeout 300 python bench_traversal.py --scale 24 --weights int --roots 1 --no-sssp --no-cpu-baseline --no-check 2>&1 | grep "\[bfs\]" | tail -22 | tee "$O/r4l_bfs_trace.log"
