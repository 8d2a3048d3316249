#!/usr/bin/env bash
# round 4, session b: IPC export limits, the communicator's self-test at more ranks / sizes, PageRank through cugraph_graph_create_mg
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
( ulimit -n; timeout 120 tools/ubench/ipc_sizes ) 2>&1 | tee "$O/r4b_ipc_sizes.log" | tail -70
for cfg in "3 65536" "4 65536" "4 4194304" "8 65536" "8 4194304"; do
  set -- $cfg; w=$1; n=$2
  d=$(mktemp -d); s="s$RANDOM"
  for r in $(seq 0 $((w-1))); do timeout 150 python tests/ipc_worker.py selftest $s $r $w $d $n 30 > $d/out$r.log 2>&1 & done; wait
  echo "== world $w n_words $n:"; cat $d/rank*.json 2>/dev/null | head -c 600; echo; for r in $(seq 0 $((w-1))); do grep -h "Error\|error" $d/out$r.log | tail -1 | cut -c1-300; done | sort | uniq -c
done 2>&1 | tee "$O/r4b_selftest.log"
timeout 900 python -m pytest tests/test_mg_capi.py -m gpu -x -q 2>&1 | tail -30 | tee "$O/r4b_mgcapi.log"
