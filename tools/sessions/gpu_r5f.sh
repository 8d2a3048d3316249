#!/usr/bin/env bash
# round 5, session f: 50 conformance runs at lease start; SSSP with the wide frontiers ordered by distance band (experiment)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 400 python tools/conformance_loop.py 50 2>&1 | tail -4 | tee "$O/r5f_conformance_loop.log"
fmt='
import sys, json
d = json.loads(sys.stdin.read())
x = d["sssp"]; print("sssp with pred", x["mean_ms"], "ms steps", x["mean_steps"], "relax/edge", x["mean_relaxations_per_edge"], "| distance only", (x.get("distance_only") or {}).get("mean_ms"), "| check", x.get("check", {}).get("ok"))'
for v in 0 1000000 32768 2048; do
  echo "== CUGRAPH_AMD_SSSP_SORT=$v" | tee -a "$O/r5f_sssp_sort.txt"
  CUGRAPH_AMD_SSSP_SORT=$v timeout 300 python bench_traversal.py --scale 24 --weights int --roots 16 --no-cpu-baseline 2>&1 | tail -1 | python -c "$fmt" | tee -a "$O/r5f_sssp_sort.txt"
done
for v in 0 32768; do
  echo "== trace, CUGRAPH_AMD_SSSP_SORT=$v" | tee -a "$O/r5f_sssp_sort.txt"
  CUGRAPH_AMD_SSSP_SORT=$v CUGRAPH_AMD_SSSP_TRACE=1 timeout 200 python bench_traversal.py --scale 24 --weights int --roots 1 --no-cpu-baseline --no-check --single-variant 2>&1 | grep "^\[sssp\]" | tail -19 | cut -c1-200 | tee -a "$O/r5f_sssp_sort.txt"
done
