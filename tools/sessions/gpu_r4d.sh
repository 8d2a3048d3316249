#!/usr/bin/env bash
# round 4, session d: arena windows at 8 ranks; the whole -m gpu suite
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in "8 4194304" "8 4194304" "6 1048576" "2 16777216"; do
  set -- $cfg; w=$1; n=$2
  d=$(mktemp -d); s="s$RANDOM"
  for r in $(seq 0 $((w-1))); do timeout 150 python tests/ipc_worker.py selftest $s $r $w $d $n 30 > $d/out$r.log 2>&1 & done; wait
  echo "== world $w n_words $n:"; cat $d/rank*.json 2>/dev/null | head -c 300; echo; for r in $(seq 0 $((w-1))); do grep -h "Error\|error" $d/out$r.log | tail -1 | cut -c1-420; done | sort | uniq -c
done 2>&1 | tee "$O/r4d_selftest.log"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee "$O/r4d_pytest_gpu.log"
