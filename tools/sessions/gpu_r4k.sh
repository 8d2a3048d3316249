#!/usr/bin/env bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_mg_capi.py -m gpu -x -q -k "louvain" 2>&1 | tail -40 | tee "$O/r4k_mg_louvain.log"
