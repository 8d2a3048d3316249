#!/usr/bin/env bash
# round 4, session z: the reference's multi-GPU C tests (7 files, unchanged) + the MG suites with the optional PageRank arguments, FLOAT64 SSSP, MG degrees, creation flags
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0 LD_LIBRARY_PATH="$R/cugraph_amd/lib:${LD_LIBRARY_PATH:-}"
for n in mg_degrees_test mg_generate_rmat_test; do
  for ranks in 2 3; do
    echo "== $n ranks=$ranks"; CUGRAPH_AMD_TEST_RANKS=$ranks timeout 120 tests/c_api/_ref_bin/$n 2>&1 | tail -25; echo "rc=$?"
  done
done 2>&1 | tee "$O/r4z_ref_mg_tests.log"
timeout 1200 python -m pytest tests/test_reference_c_tests.py tests/test_mg_capi.py tests/test_c_conformance.py tests/test_pylibcugraph_on_gpu.py -m gpu -x -q 2>&1 | tail -12 | tee "$O/r4z_suites.log"
