#!/usr/bin/env bash
# GPU box: everything the round's numbers come from, in one session: -m gpu suite, smoke, bench line, Louvain line, traversal lines + kernel summary
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
TAG=${TAG:-r2k}
timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} 2>&1 | tail -6 | tee "$O/pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$O/smoke.log"
timeout 600 python bench.py 2>"$O/bench.err" | tee "$O/${TAG}_bench_s26.json" | cut -c1-400
timeout 600 python bench_louvain.py --scale 22 --out "$O/${TAG}_louvain_s22.json" 2>/dev/null | cut -c1-700
TAG=$TAG PROF=${PROF:-1} bash tools/gpu_trav.sh 2>&1 | cut -c1-300 | tail -24
