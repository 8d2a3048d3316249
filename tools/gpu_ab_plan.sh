#!/usr/bin/env bash
# GPU box: same-session A/B of library variants (gpurun_libs/<name>.so, LIBS="a b c") on the PageRank plan: REPS interleaved rounds of
# tools/plan_sweep.py at SCALE (one fresh process per measurement); boxes differ by +-4 % in phase 1, so only same-session pairs compare.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
cp cugraph_amd/lib/libcugraph_c.so /tmp/orig.so
: > "$O/ab_plan.log"
for rep in $(seq 1 ${REPS:-3}); do for lib in ${LIBS:-old}; do
  cp "gpurun_libs/$lib.so" cugraph_amd/lib/libcugraph_c.so
  timeout 200 python tools/plan_sweep.py --scale ${SCALE:-26} --steps ${STEPS:-20} --reps 1 ${VARIANTS:-base} 2>&1 | grep "^rep" | sed "s/^rep 0/rep $rep lib=$lib/" | tee -a "$O/ab_plan.log"
done; done
cp /tmp/orig.so cugraph_amd/lib/libcugraph_c.so
