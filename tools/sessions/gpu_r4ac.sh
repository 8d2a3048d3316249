#!/usr/bin/env bash
# round 4, session ac: the multi-GPU suites with the communicator's windows in FINE-GRAINED memory (what ranks on different GPUs get by default)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for kind in finegrained uncached; do
  echo "== CUGRAPH_AMD_COMM_WINDOWS=$kind"
  CUGRAPH_AMD_COMM_WINDOWS=$kind timeout 1200 python -m pytest tests/test_mg_capi.py tests/test_reference_c_tests.py tests/test_c_conformance.py tests/test_pylibcugraph_on_gpu.py -m gpu -q -k "mg or MG or comm or rank" 2>&1 | tail -4
done 2>&1 | tee "$O/r4ac_window_kinds.log"
