#!/usr/bin/env bash
# GPU box, round 3 session C: phase-2 A/B (old loop / double-buffered / double-buffered + late epilogue inputs), exit-crash hunt
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
LIBS="p2old p2new p2late p2old p2new p2late" BENCH_EXTRA="--no-extras" bash tools/gpu_ab.sh 2>&1 | tail -20
cp "$O/ab.log" "$O/r3c_ab.log"
timeout 600 python -X faulthandler bench.py --steps 5 --warmup 2 --cpu-scale 18 > "$O/r3c_exit.json" 2> "$O/r3c_exit.err"; echo "bench rc=$?" | tee -a "$O/r3c_exit.err"
tail -30 "$O/r3c_exit.err" | cut -c1-300
cut -c1-200 "$O/r3c_exit.json"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pagerank and not config2 and not full_size" 2>&1 | tail -4
