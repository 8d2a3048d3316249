#!/usr/bin/env bash
# round 6, session b: phase 2 made lean (no overlap / batch / shift variants, tile scale from the exponent field, epilogue stores after all rows): PageRank
# parity tests, then base (HEAD of round 5) against the new library, interleaved
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pagerank" 2>&1 | tail -5 | tee "$O/r6b_tests.log"
cp cugraph_amd/lib/libcugraph_c.so gpurun_libs/new.so
LIBS="${LIBS:-base new}" REPS=3 bash tools/gpu_ab_plan.sh
cp "$O/ab_plan.log" "$O/${TAG:-r6b}_ab.txt"
