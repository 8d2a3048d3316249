#!/usr/bin/env bash
# GPU box: parameter sweeps of the traversals' host heuristics on bench_traversal.py (RMAT-24, 32 roots): SSSP bucket width, BFS direction thresholds
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
out="$O/${TAG:-r6av}_traversal_knobs.txt"; : > "$out"
for ds in ${DS:-1 4 16 64 1024}; do
  echo -n "SSSP_DELTA_SCALE=$ds: " | tee -a "$out"
  CUGRAPH_AMD_SSSP_DELTA_SCALE=$ds timeout 300 python bench_traversal.py --scale 24 --weights int --roots 32 --no-cpu-baseline 2>/dev/null | python tools/trav_line.py | tee -a "$out"
done
for ab in "60 24" "240 24" "480 24" "1000 24" "240 48" "60 24" "240 24"; do
  set -- $ab
  echo -n "BFS_ALPHA=$1 BFS_BETA=$2: " | tee -a "$out"
  CUGRAPH_AMD_BFS_ALPHA=$1 CUGRAPH_AMD_BFS_BETA=$2 timeout 300 python bench_traversal.py --scale 24 --weights int --roots 32 --no-cpu-baseline --no-sssp 2>/dev/null | python tools/trav_line.py | tee -a "$out"
done
