#!/usr/bin/env bash
# GPU box, round 3 session D: phase-2 fixed-point conversion A/B + PageRank parity
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pagerank and not full_size" 2>&1 | tail -4
LIBS="p2oldfix p2fix p2fixlate p2oldfix p2fix p2fixlate" BENCH_EXTRA="--no-extras" bash tools/gpu_ab.sh 2>&1 | tail -14
cp "$O/ab.log" "$O/r3d_ab.log"
