#!/usr/bin/env bash
# GPU box, round 3 session K: full -m gpu suite, Louvain RMAT-26 with the pool cap restored
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -16 | tee "$O/r3k_pytest.log"
CUGRAPH_AMD_LOUVAIN_TRACE=1 timeout 600 python bench_louvain.py --scale 26 --cpu-scale 0 --repeats 2 --out "$O/r3k_louvain_s26.json" 2>&1 | grep -E "\[louvain\] [0-9]|\"value\"" | tail -4 | cut -c1-200
