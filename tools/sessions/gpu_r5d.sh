#!/usr/bin/env bash
# round 5, session d: BFS parents pulled per discovered vertex after the long-row fix (parity, then A/B of the timing)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 python tools/debug/bfs_pull_debug.py 2>&1 | grep "^scale\|^   v" | tee "$O/r5d_bfs_pull_debug.log"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sssp or bfs or extract_paths" --durations=5 2>&1 | tail -12 | tee "$O/r5d_tests.log"
fmt='
import sys, json
d = json.loads(sys.stdin.read())
for k in ("bfs", "sssp"):
    if k in d:
        x = d[k]; print(k, "with pred", x["mean_ms"], "ms (min", x["min_ms"], "max", x["max_ms"], ") frac", x["roofline"]["frac"], "| distance only", (x.get("distance_only") or {}).get("mean_ms"), "| check", x.get("check", {}).get("ok"))'
echo "== BFS parents pulled" | tee "$O/r5d_bfs_ab.txt"
timeout 300 python bench_traversal.py --scale 24 --weights int --roots 32 --no-sssp --no-cpu-baseline 2>&1 | tail -1 | python -c "$fmt" | tee -a "$O/r5d_bfs_ab.txt"
echo "== BFS parents by atomicMin in the push" | tee -a "$O/r5d_bfs_ab.txt"
CUGRAPH_AMD_BFS_PULL_PARENTS=0 timeout 300 python bench_traversal.py --scale 24 --weights int --roots 32 --no-sssp --no-cpu-baseline --no-check 2>&1 | tail -1 | python -c "$fmt" | tee -a "$O/r5d_bfs_ab.txt"
echo "== symmetrised, pulled" | tee -a "$O/r5d_bfs_ab.txt"
timeout 300 python bench_traversal.py --scale 24 --symmetric --roots 16 --no-sssp --no-cpu-baseline 2>&1 | tail -1 | python -c "$fmt" | tee -a "$O/r5d_bfs_ab.txt"
CUGRAPH_AMD_BFS_TRACE=1 timeout 200 python bench_traversal.py --scale 24 --weights int --roots 2 --no-sssp --no-cpu-baseline --no-check --single-variant 2>&1 | grep "^\[bfs\]" | tail -13 | tee "$O/r5d_bfs_trace.log"
