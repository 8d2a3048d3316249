#!/usr/bin/env bash
# round 4, session a: does the library's communicator work with several processes on one GPU (HIP IPC, flags, peer pushes)?
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_mg_capi.py -m gpu -x -q 2>&1 | tail -15 | tee "$O/r4a_comm.log"
for w in 2 4 8; do
  d=$(mktemp -d); s="s$RANDOM"
  for r in $(seq 0 $((w-1))); do timeout 120 python tests/ipc_worker.py selftest $s $r $w $d $((1<<22)) 50 > $d/out$r.log 2>&1 & done; wait
  echo "world $w:"; cat $d/rank*.json 2>/dev/null | head -c 1500; echo; tail -3 $d/out0.log
done 2>&1 | tee -a "$O/r4a_comm.log"
timeout 600 python -m pytest tests/test_mg.py -m gpu -x -q -k "2d_hip_engine" 2>&1 | tail -5 | tee -a "$O/r4a_comm.log"
