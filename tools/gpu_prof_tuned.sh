#!/usr/bin/env bash
# GPU box: rocprofv3 kernel trace of the DRIVER's command line (tuned plan, 20 steps, 5 warm-up) without the phase pass: the last 20 launches of
# k_tiled_phase1 / k_tiled_phase2 in the trace are the timed region (rocpd_summary.py --last 20 k_tiled_phase); the bench line of the same process beside it
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; TAG=${TAG:-r6z}
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/prof_${TAG}_tuned"
timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_${TAG}_tuned" -o run -- python "$R/bench.py" --gpus 1 --steps 20 --warmup 5 --no-check --no-cpu-baseline --no-extras --no-phase-pass > "$O/${TAG}_s26_tuned_under_rocprof.json" 2> "$O/${TAG}_s26_tuned_under_rocprof.err"
python "$R/tools/rocpd_summary.py" --last 20 k_tiled_phase "$O/prof_${TAG}_tuned" > "$O/${TAG}_s26_tuned_rocprofv3_summary.txt" 2>&1
find "$O/prof_${TAG}_tuned" -name "*.db" -delete
grep "^{" "$O/${TAG}_s26_tuned_under_rocprof.json" | cut -c1-200
grep -A4 "^# last" "$O/${TAG}_s26_tuned_rocprofv3_summary.txt" | cut -c1-150
