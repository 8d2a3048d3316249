#!/usr/bin/env bash
# round 4, session o: SSSP pull rounds: parity (forced pull), then the RMAT-24 line with and without them, and the round trace
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sssp" 2>&1 | tail -5 | tee "$O/r4o_sssp_tests.log"
for pull in 1 0; do
  CUGRAPH_AMD_SSSP_PULL=$pull timeout 600 python bench_traversal.py --scale 24 --weights int --roots 16 --no-cpu-baseline --out "$O/r4o_s24_int_pull$pull.json" > /dev/null 2>&1
  CUGRAPH_AMD_SSSP_PULL=$pull timeout 600 python bench_traversal.py --scale 24 --weights unit --roots 16 --no-cpu-baseline --out "$O/r4o_s24_unit_pull$pull.json" > /dev/null 2>&1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4o_s24_*.json")):
    d=json.load(open(f)); s=d["sssp"]
    print(f.split("/")[-1], "sssp mean_ms", s.get("mean_ms"), "min/max", s.get("min_ms"), s.get("max_ms"), "frac", s["roofline"]["frac"], "relax/edge", s.get("mean_relaxations_per_edge"), "check", (s.get("check") or {}).get("ok"), "| bfs", d["bfs"]["mean_ms"])
PY
CUGRAPH_AMD_SSSP_TRACE=1 timeout 300 python bench_traversal.py --scale 24 --weights int --roots 1 --no-cpu-baseline --no-check 2>&1 | grep "\[sssp\]" | tail -20 | cut -c1-230 | tee "$O/r4o_sssp_trace.log"
