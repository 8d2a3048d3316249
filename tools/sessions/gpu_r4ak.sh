#!/usr/bin/env bash
# round 4, session ak: the multi-GPU suites after the windows became opt-in
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_mg_capi.py tests/test_reference_c_tests.py tests/test_c_conformance.py tests/test_pylibcugraph_on_gpu.py tests/test_mg_traversal.py -m gpu -x -q 2>&1 | tail -4 | tee "$O/r4ak_tests.log"
