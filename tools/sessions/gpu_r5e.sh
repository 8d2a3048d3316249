#!/usr/bin/env bash
# round 5, session e: slot blocks of the partial buffer aligned to cache lines (16 / 32 / 64 slots) -- parity, then phase 1 / phase 2 times
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "aligned_slot or phase1_schedules or pulled_parents" --durations=3 2>&1 | tail -8 | tee "$O/r5e_tests.log"
A=CUGRAPH_AMD_TILED_BLOCK_ALIGN
CUGRAPH_AMD_TILED_DEBUG=1 timeout 200 python tools/plan_sweep.py --scale 26 --steps 20 --reps 3 base $A=16 $A=32 $A=64 2>&1 | grep "^rep\|tiled build\] T" | tee "$O/r5e_align_s26.log"
timeout 100 python tools/plan_sweep.py --scale 24 --steps 40 --reps 2 base $A=16 $A=32 2>&1 | grep "^rep" | tee "$O/r5e_align_s24.log"
timeout 100 python tools/plan_sweep.py --scale 22 --steps 100 --reps 2 base $A=16 $A=32 2>&1 | grep "^rep" | tee "$O/r5e_align_s22.log"
