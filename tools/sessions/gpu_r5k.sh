#!/usr/bin/env bash
# round 5, session k: the tiled plan with 64-bit per-wavefront bases -- unchanged speed at RMAT-26?  parity; then RMAT-27 (2^31 edges) through bench.py
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pagerank" --durations=3 2>&1 | tail -6 | tee "$O/r5k_tests.log"
timeout 120 python tools/plan_sweep.py --scale 26 --steps 20 --reps 3 base 2>&1 | grep "^rep" | tee "$O/r5k_s26.log"
timeout 100 python tools/plan_sweep.py --scale 24 --steps 40 --reps 2 base 2>&1 | grep "^rep" | tee -a "$O/r5k_s26.log"
timeout 100 python tools/plan_sweep.py --scale 22 --steps 100 --reps 2 base 2>&1 | grep "^rep" | tee -a "$O/r5k_s26.log"
timeout 600 python bench.py --scale 27 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > "$O/r5k_bench_s27.json" 2> "$O/r5k_bench_s27.err"; echo "rc=$?"; tail -3 "$O/r5k_bench_s27.err" | cut -c1-400; cut -c1-900 "$O/r5k_bench_s27.json"
