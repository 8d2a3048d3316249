#!/usr/bin/env bash
# GPU box: the -m gpu suite, smoke, and the default bench line (what the driver runs at round end), logs in gpurun_out/
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
timeout 1500 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} 2>&1 | tail -15 | tee "$O/pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$O/smoke.log"
CUGRAPH_AMD_TILED_DEBUG=${DBG:-} timeout 600 python bench.py ${BENCH_ARGS:-} 2>"$O/bench.err" | tee "$O/bench.json"
grep "tiled" "$O/bench.err" | head
