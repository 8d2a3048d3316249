#!/usr/bin/env bash
# GPU box: Louvain tests + kernel profile of bench_louvain.py, build trace of bench.py; logs in gpurun_out/
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q -k "${PYTEST_K:-louvain or histogram or degrees}" 2>&1 | tail -8 | tee "$O/pytest_b.log"
P="$O/prof_louvain"; rm -rf "$P"; mkdir -p "$P"
( cd /tmp && export TMPDIR=/tmp && CUGRAPH_AMD_LOUVAIN_TRACE=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$P/stats" -o run -- python $R/bench_louvain.py --scale ${LV_SCALE:-22} --repeats 1 --cpu-scale 0 > "$P/stats.log" 2>&1 )
python tools/rocpd_summary.py "$P" > "$P/summary.txt" 2>&1
find "$P" -name "*.db" -delete
head -32 "$P/summary.txt" | cut -c1-150
grep "louvain\]" "$P/stats.log" | tail -4
CUGRAPH_AMD_BUILD_TRACE=1 timeout 600 python bench.py --steps 5 --no-cpu-baseline --no-check 2>"$O/bench_trace.err" | cut -c1-300
grep "\[build\]" "$O/bench_trace.err"
