#!/usr/bin/env bash
# round 5, session l: scalar bases + constant lane offsets for every phase-1 load (bitmap, records, deltas): parity and timing
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pagerank" 2>&1 | tail -3 | tee "$O/r5l_tests.log"
timeout 120 python tools/plan_sweep.py --scale 26 --steps 20 --reps 3 base 2>&1 | grep "^rep" | tee "$O/r5l_s26.log"
timeout 100 python tools/plan_sweep.py --scale 24 --steps 40 --reps 2 base 2>&1 | grep "^rep" | tee -a "$O/r5l_s26.log"
timeout 100 python tools/plan_sweep.py --scale 22 --steps 100 --reps 2 base 2>&1 | grep "^rep" | tee -a "$O/r5l_s26.log"
