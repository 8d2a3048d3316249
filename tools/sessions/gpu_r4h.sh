#!/usr/bin/env bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_c_conformance.py tests/test_pylibcugraph_on_gpu.py -m gpu -x -q 2>&1 | tail -40 | tee "$O/r4h_n3.log"
