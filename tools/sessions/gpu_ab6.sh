#!/usr/bin/env bash
# round 6: same-session A/B of gpurun_libs/<name>.so on the PageRank plan (LIBS, REPS, SCALE, TAG); the tree's library is added as "new"
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
cp cugraph_amd/lib/libcugraph_c.so gpurun_libs/new.so
bash tools/gpu_ab_plan.sh
cp "$O/ab_plan.log" "$O/${TAG:-r6}_ab.txt"
