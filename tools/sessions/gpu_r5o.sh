#!/usr/bin/env bash
# round 5, session o: phase 1 with absolute-address LDS gathers (no `v_add 0` per gather) and with the branch-free staging (dump word):
# parity of both variants, then a same-session A/B against the generic-pointer kernel
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
cp cugraph_amd/lib/libcugraph_c.so /tmp/orig.so
for lib in abs absdump; do
  cp "gpurun_libs/$lib.so" cugraph_amd/lib/libcugraph_c.so
  echo "== parity with lib=$lib" | tee -a "$O/r5o_tests.log"
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pagerank and not two_to_the" 2>&1 | tail -3 | tee -a "$O/r5o_tests.log"
done
cp /tmp/orig.so cugraph_amd/lib/libcugraph_c.so
LIBS="generic abs absdump" REPS=3 bash tools/gpu_ab_plan.sh
cp "$O/ab_plan.log" "$O/r5o_ab_s26.log"
LIBS="generic abs absdump" REPS=2 SCALE=22 STEPS=100 bash tools/gpu_ab_plan.sh
cp "$O/ab_plan.log" "$O/r5o_ab_s22.log"
