#!/usr/bin/env bash
# round 5, session n: SSSP with predecessors, 4-byte pre-test + packed parent word (hybrid) -- parity and timing
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sssp" 2>&1 | tail -4 | tee "$O/r5n_tests.log"
fmt='
import sys, json
d = json.loads(sys.stdin.read())
x = d["sssp"]; print("sssp with pred", x["mean_ms"], "ms steps", x["mean_steps"], "relax/edge", x["mean_relaxations_per_edge"], "| distance only", (x.get("distance_only") or {}).get("mean_ms"), "| check", x.get("check", {}).get("ok"))'
for w in int int unit; do
  echo "== weights $w" | tee -a "$O/r5n_sssp.txt"
  timeout 300 python bench_traversal.py --scale 24 --weights $w --roots 16 --no-cpu-baseline 2>&1 | tail -1 | python -c "$fmt" | tee -a "$O/r5n_sssp.txt"
done
