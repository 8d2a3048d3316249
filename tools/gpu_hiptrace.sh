#!/usr/bin/env bash
# GPU box: HIP API time breakdown (hipMalloc / hipFree / memcpy / sync) of bench.py -- where host time of graph / plan build goes
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; P="$O/hiptrace"; rm -rf "$P"; mkdir -p "$P"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --hip-trace --stats -f csv -d "$P" -o run -- python $R/bench.py --scale ${SCALE:-26} --steps 3 --warmup 1 --no-cpu-baseline > "$P/log.txt" 2>&1
ls "$P"; for f in "$P"/*hip_api_stats.csv "$P"/*/*hip_api_stats.csv; do [ -f "$f" ] && head -15 "$f"; done; tail -2 "$P/log.txt" | cut -c1-300
