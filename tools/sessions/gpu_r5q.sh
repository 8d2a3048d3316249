#!/usr/bin/env bash
# GPU box, round 5 session q: the multi-GPU graph at the width of the single-GPU one (INT64 ids, edge properties, decompress, extract_paths) --
# the new tests first, then everything that shares code with them
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_mg_capi.py -m gpu -q -k "int64 or edge_properties or extract_paths" 2>&1 | tail -60 | tee "$O/r5q_new.log"
timeout 900 python -m pytest tests/test_mg_capi.py tests/test_reference_c_tests.py -m gpu -q -k "not int64 and not edge_properties and not extract_paths and not rmat22" 2>&1 | tail -15 | tee "$O/r5q_mg.log"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "int64 or outer or sparse or edge_ids or decompress or extract or degrees or has_vertex or goldens" 2>&1 | tail -8 | tee "$O/r5q_parity.log"
