#!/usr/bin/env bash
# GPU box: Louvain after a change -- parity tests, RMAT-22 per-level trace, kernel summary
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${TAG:-r6w}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "louvain" 2>&1 | tail -5 | tee "$O/${TAG}_pytest_louvain.log"
CUGRAPH_AMD_LOUVAIN_TRACE=1 timeout 600 python bench_louvain.py --scale 22 --cpu-scale 0 --repeats 3 > "$O/${TAG}_louvain_trace.txt" 2>&1
grep -E "louvain\]" "$O/${TAG}_louvain_trace.txt" | tail -7; grep '^{' "$O/${TAG}_louvain_trace.txt" | cut -c1-400
( cd /tmp && export TMPDIR=/tmp
rm -rf "$O/prof_$TAG"; timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_$TAG/louv" -o run -- python "$R/bench_louvain.py" --scale 22 --cpu-scale 0 --repeats 3 > "$O/prof_$TAG.log" 2>&1
python "$R/tools/rocpd_summary.py" "$O/prof_$TAG/louv" > "$O/${TAG}_louvain_s22_rocprofv3_summary.txt" 2>&1
python "$R/tools/rocpd_summary.py" --segments k_lv_chunk_prep "$O/prof_$TAG/louv" > "$O/${TAG}_louvain_s22_levels.txt" 2>&1
find "$O/prof_$TAG" -name "*.db" -delete
head -24 "$O/${TAG}_louvain_s22_rocprofv3_summary.txt" | cut -c1-150; cut -c1-600 "$O/${TAG}_louvain_s22_levels.txt" )
