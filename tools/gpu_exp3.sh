#!/usr/bin/env bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q -k "pagerank or mg" 2>&1 | tail -3
fmt='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print("ms/step", d["ms_per_step"], "p1", r.get("avg_phase1_ms"), "p2", r.get("avg_phase2_ms"), "frac", r["frac"], "check", (d.get("check") or {}).get("ok"))'
for rep in 1 2; do
for v in "A=0" "CUGRAPH_AMD_TILED_DSTL16=1"; do
  echo "== $v"; env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-check 2>&1 | python -c "$fmt"
done; done
