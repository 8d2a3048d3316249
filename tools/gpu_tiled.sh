#!/usr/bin/env bash
# GPU box: PageRank tests first, then tiled-vs-flat at the requested scales and tile sizes.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${TESTS:-pagerank or golden or karate or empty}" > "$O/pytest_pr.log" 2>&1; echo "pytest_rc=$?" >> "$O/pytest_pr.log"
tail -15 "$O/pytest_pr.log"
: > "$O/tiled.log"
fmt='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print(d["ms_per_step"], d["value"], r["avg_kernel_ms"], r.get("avg_phase1_ms"), r.get("avg_phase2_ms"), r["frac"], d["graph_build_s"], d.get("plan_build_s"))'
for scale in ${SCALES:-22 26}; do for cfg in ${CFGS:-tiled:0 tiled:16384 flat:16384}; do
  kern=${cfg%%:*}; hot=${cfg##*:}
  echo "== scale=$scale kernel=$kern tile=$hot" >> "$O/tiled.log"
  CUGRAPH_AMD_PAGERANK_KERNEL=$kern timeout 600 python bench.py --scale $scale --steps 20 --warmup 3 --no-cpu-baseline --hot-tile $hot 2>&1 | tail -3 | python -c "$fmt" >> "$O/tiled.log" 2>&1
done; done
cat "$O/tiled.log"
