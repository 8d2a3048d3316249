#!/usr/bin/env bash
# GPU box: kernel-trace stats of bench_traversal.py (BFS + SSSP) at one scale; summary into gpurun_out/prof_trav_<tag>/
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"
SCALE="${SCALE:-24}"; TAG="${TAG:-r1}"
P="$O/prof_trav_$TAG"; rm -rf "$P"; mkdir -p "$P"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$P/stats" -o run -- python $R/bench_traversal.py --scale $SCALE --roots ${ROOTS:-4} --weights ${WEIGHTS:-int} ${EXTRA:-} > "$P/stats.log" 2>&1
python "$R/tools/rocpd_summary.py" "$P" > "$P/summary.txt" 2>&1
find "$P" -name "*.db" -delete
head -40 "$P/summary.txt"; tail -2 "$P/stats.log"
