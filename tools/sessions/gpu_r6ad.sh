#!/usr/bin/env bash
# round 6: phase-1 schedule knobs at the small scales (RMAT-22 / 23: BASELINE config 2 and a rank's share at 8 GPUs), interleaved on one graph
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${TAG:-r6ad}
{
for sc in 22 23; do
echo "=== scale $sc"
timeout 600 python tools/plan_sweep.py --scale $sc --steps 50 --reps 3 base CUGRAPH_AMD_TP_STATIC_FRAC=0.5 CUGRAPH_AMD_TP_STATIC_FRAC=0.7 CUGRAPH_AMD_TP_STATIC_FRAC=0.8 CUGRAPH_AMD_TP_STATIC_FRAC=0.95 CUGRAPH_AMD_TP_STATIC_FRAC=1.0 CUGRAPH_AMD_TP_STATIC_FRAC=0 2>&1 | tail -30
done
} 2>&1 | tee "$O/${TAG}_small_scale_schedule.txt"
