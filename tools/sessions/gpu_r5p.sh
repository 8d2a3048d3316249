#!/usr/bin/env bash
# round 5, session p: BFS -- smaller deferred work units for narrow frontiers (parity, then the unit size swept)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bfs" 2>&1 | tail -3 | tee "$O/r5p_tests.log"
fmt='
import sys, json
d = json.loads(sys.stdin.read())
x = d["bfs"]; print("bfs with pred", x["mean_ms"], "ms (min", x["min_ms"], "max", x["max_ms"], ") | distance only", (x.get("distance_only") or {}).get("mean_ms"), "| check", x.get("check", {}).get("ok"))'
for v in 4096 512 1024 256 4096 512; do
  echo "== CUGRAPH_AMD_BFS_NARROW_SEG=$v" | tee -a "$O/r5p_bfs_seg.txt"
  CUGRAPH_AMD_BFS_NARROW_SEG=$v timeout 300 python bench_traversal.py --scale 24 --weights int --roots 32 --no-sssp --no-cpu-baseline 2>&1 | tail -1 | python -c "$fmt" | tee -a "$O/r5p_bfs_seg.txt"
done
