#!/usr/bin/env bash
# GPU box: (1) placement trials by group (pagerank.hip: tune -- A = phase 1's read streams, B = the partial buffer, C = phase 2's destinations), fresh processes of the
# driver's command line with 8 / 14 placements, the per-phase trace of every trial; (2) Louvain RMAT-22 per-level trace (where the 70 ms go)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > "$O/r6v_placement_groups.txt"
for rep in 1 2 3 4; do for n in 8 14; do
  CUGRAPH_AMD_PR_PLACEMENT_TRACE=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --placements $n --no-extras --no-cpu-baseline --no-check 2>&1 | grep -E "placement|ms_per_step" \
   | sed -E "s/.*\"ms_per_step\": ([0-9.]+).*avg_phase1_ms\": ([0-9.]+), \"avg_phase2_ms\": ([0-9.]+).*/rep $rep trials $n: bench ms_per_step \1 phase1 \2 phase2 \3/" | cut -c1-260 >> "$O/r6v_placement_groups.txt"
done; done
grep "^rep" "$O/r6v_placement_groups.txt"
CUGRAPH_AMD_LOUVAIN_TRACE=1 timeout 600 python bench_louvain.py --scale 22 --cpu-scale 0 --repeats 2 > "$O/r6v_louvain_trace.txt" 2>&1
grep -E "louvain\]" "$O/r6v_louvain_trace.txt" | tail -14
