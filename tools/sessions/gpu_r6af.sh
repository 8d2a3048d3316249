#!/usr/bin/env bash
# round 6: Louvain on the regenerated RMAT-22 fixture (single GPU, 2 ranks), its timing and kernel profile with the reference's level numbering
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${TAG:-r6af}
timeout 900 python -m pytest tests -m gpu -q -k "rmat_golden and 22 or rmat22_golden" 2>&1 | tail -5 | tee "$O/${TAG}_pytest_louvain_s22.log"
CUGRAPH_AMD_LOUVAIN_TRACE=1 timeout 600 python bench_louvain.py --scale 22 --cpu-scale 0 --repeats 3 --out "$O/${TAG}_louvain_s22.json" 2>&1 | grep "louvain\]" | tail -8 | tee "$O/${TAG}_louvain_s22_levels.txt"
cut -c1-400 "$O/${TAG}_louvain_s22.json"
( cd /tmp && export TMPDIR=/tmp
rm -rf "$O/prof_$TAG"; timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_$TAG/louv" -o run -- python "$R/bench_louvain.py" --scale 22 --cpu-scale 0 --repeats 3 > "$O/prof_$TAG.log" 2>&1
python "$R/tools/rocpd_summary.py" "$O/prof_$TAG/louv" > "$O/${TAG}_louvain_s22_rocprofv3_summary.txt" 2>&1
find "$O/prof_$TAG" -name "*.db" -delete
head -14 "$O/${TAG}_louvain_s22_rocprofv3_summary.txt" | cut -c1-150 )
