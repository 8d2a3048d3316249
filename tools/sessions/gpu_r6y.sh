#!/usr/bin/env bash
# GPU box: is the Louvain chunk kernel bound by the lines its gathers miss?  The same RMAT-22 graph with its vertices numbered by descending degree (hot
# destinations share cache lines) against the generator's numbering: per-level trace + kernel summary of both
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${TAG:-r6y}
for mode in none degree; do
( cd /tmp && export TMPDIR=/tmp
rm -rf "$O/prof_$TAG"; BENCH_LOUVAIN_RELABEL=$mode CUGRAPH_AMD_LOUVAIN_TRACE=1 timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_$TAG/louv" -o run -- python "$R/bench_louvain.py" --scale 22 --cpu-scale 0 --repeats 2 > "$O/prof_$TAG.log" 2>&1
grep -E "louvain\]" "$O/prof_$TAG.log" | tail -6 > "$O/${TAG}_relabel_$mode.txt"
python "$R/tools/rocpd_summary.py" "$O/prof_$TAG/louv" | head -12 | cut -c1-150 >> "$O/${TAG}_relabel_$mode.txt" 2>&1
python "$R/tools/rocpd_summary.py" --segments k_lv_chunk_prep "$O/prof_$TAG/louv" | cut -c1-700 >> "$O/${TAG}_relabel_$mode.txt" 2>&1
find "$O/prof_$TAG" -name "*.db" -delete
cat "$O/${TAG}_relabel_$mode.txt" )
done
