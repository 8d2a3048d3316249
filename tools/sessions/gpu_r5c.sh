#!/usr/bin/env bash
# round 5, session c: SSSP with the parent packed into the relaxation, BFS parents pulled per discovered vertex (A/B against the
# sweep / the atomicMin in the push), phase 2's two fixed-point conversions side by side, the MG PageRank tests after the channel fix.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sssp or bfs or extract_paths" --durations=5 2>&1 | tail -12 | tee "$O/r5c_tests.log"
timeout 600 python -m pytest tests/test_mg_capi.py -m gpu -x -q -k "pagerank or selftest" 2>&1 | tail -5 | tee -a "$O/r5c_tests.log"
fmt='
import sys, json
d = json.loads(sys.stdin.read())
for k in ("bfs", "sssp"):
    x = d[k]; print(k, "with pred", x["mean_ms"], "ms frac", x["roofline"]["frac"], "| distance only", (x.get("distance_only") or {}).get("mean_ms"), "| check", x.get("check", {}).get("ok"))'
echo "== new (packed SSSP parents, pulled BFS parents)" | tee "$O/r5c_traversal_ab.txt"
timeout 300 python bench_traversal.py --scale 24 --weights int --roots 16 --no-cpu-baseline --out "$O/r5c_traversal_s24_int.json" 2>&1 | tail -1 | python -c "$fmt" | tee -a "$O/r5c_traversal_ab.txt"
echo "== old (parent sweep, atomicMin in the push)" | tee -a "$O/r5c_traversal_ab.txt"
CUGRAPH_AMD_SSSP_PACKED=0 CUGRAPH_AMD_BFS_PULL_PARENTS=0 timeout 300 python bench_traversal.py --scale 24 --weights int --roots 16 --no-cpu-baseline --no-check 2>&1 | tail -1 | python -c "$fmt" | tee -a "$O/r5c_traversal_ab.txt"
echo "== unit weights, new" | tee -a "$O/r5c_traversal_ab.txt"
timeout 300 python bench_traversal.py --scale 24 --weights unit --roots 16 --no-cpu-baseline --out "$O/r5c_traversal_s24_unit.json" 2>&1 | tail -1 | python -c "$fmt" | tee -a "$O/r5c_traversal_ab.txt"
timeout 120 python tools/plan_sweep.py --scale 26 --steps 20 --reps 3 base CUGRAPH_AMD_P2_FIXED=shift 2>&1 | grep "^rep" | tee "$O/r5c_p2_fixed_ab.log"
CUGRAPH_AMD_BFS_TRACE=1 timeout 200 python bench_traversal.py --scale 24 --weights int --roots 2 --no-sssp --no-cpu-baseline --no-check --single-variant 2>&1 | grep "^\[bfs\]" | tail -14 | tee "$O/r5c_bfs_trace.log"
