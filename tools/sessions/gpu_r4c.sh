#!/usr/bin/env bash
# round 4, session c: the communicator at 8 ranks / large windows with cached vs uncached windows; PageRank through cugraph_graph_create_mg
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for kind in uncached cached; do for cfg in "8 4194304" "8 4194304" "6 1048576"; do
  set -- $cfg; w=$1; n=$2
  d=$(mktemp -d); s="s$RANDOM"
  for r in $(seq 0 $((w-1))); do CUGRAPH_AMD_COMM_WINDOWS=$kind timeout 150 python tests/ipc_worker.py selftest $s $r $w $d $n 30 > $d/out$r.log 2>&1 & done; wait
  echo "== $kind world $w n_words $n:"; cat $d/rank*.json 2>/dev/null | head -c 300; echo; for r in $(seq 0 $((w-1))); do grep -h "Error\|error" $d/out$r.log | tail -1 | cut -c1-420; done | sort | uniq -c
done; done 2>&1 | tee "$O/r4c_selftest.log"
timeout 900 python -m pytest tests/test_mg_capi.py -m gpu -x -q 2>&1 | tail -30 | tee "$O/r4c_mgcapi.log"
