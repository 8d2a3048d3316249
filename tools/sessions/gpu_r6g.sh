#!/usr/bin/env bash
# round 6, session g: SSSP with the L2-resident distance filter: parity tests of everything traversal, then RMAT-24 integer weights, filter on / off in one session,
# and one traced traversal each (probes per round)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sssp or bfs" 2>&1 | tail -5 | tee "$O/r6g_tests.log"
fmt='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        j=json.loads(l); s=j["sssp"]; b=j["bfs"]; print("sssp mean_ms", s["mean_ms"], "frac", s["roofline"]["frac"], "relax/edge", s.get("relaxations_per_edge"), "dist-only", (s.get("distance_only") or {}).get("mean_ms"), "| bfs", b["mean_ms"], "check", (j.get("check") or {}).get("ok"))'
for rep in 1 2; do for f in 1 0; do for w in int unit; do
  echo -n "rep $rep filter=$f weights=$w: "
  CUGRAPH_AMD_SSSP_FILTER=$f timeout 300 python bench_traversal.py --scale 24 --roots ${ROOTS:-16} --weights $w --no-cpu-baseline 2>/dev/null | python -c "$fmt"
done; done; done | tee "$O/r6g_sssp_filter_ab.txt"
for f in 1 0; do echo "== filter=$f"; CUGRAPH_AMD_SSSP_FILTER=$f CUGRAPH_AMD_SSSP_TRACE=1 timeout 300 python bench_traversal.py --scale 24 --roots 1 --weights int --no-cpu-baseline --no-check --single-variant 2>&1 | grep "^\[sssp\]" | tail -22; done > "$O/r6g_sssp_trace.txt"
tail -44 "$O/r6g_sssp_trace.txt" | cut -c1-230
