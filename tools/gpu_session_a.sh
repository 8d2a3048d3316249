#!/usr/bin/env bash
# GPU box: -m gpu suite, Louvain timing line (trace on stderr), default bench line; logs in gpurun_out/
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
timeout 1500 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} 2>&1 | tail -15 | tee "$O/pytest_gpu.log"
CUGRAPH_AMD_LOUVAIN_TRACE=1 timeout 600 python bench_louvain.py --scale ${LV_SCALE:-22} --out "$O/louvain_s${LV_SCALE:-22}.json" 2>"$O/louvain.err" | cut -c1-1500
grep "louvain" "$O/louvain.err" | tail -30
timeout 600 python bench.py ${BENCH_ARGS:-} 2>"$O/bench.err" | tee "$O/bench.json" | cut -c1-600
tail -3 "$O/bench.err"
