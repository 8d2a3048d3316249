#!/usr/bin/env bash
# GPU box, round 3 session E: phase 2 against the round-2 code in ONE session, Louvain hash path (parity + time), BFS trace
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "louvain" 2>&1 | tail -4
for mode in 1 0; do
  CUGRAPH_AMD_LOUVAIN_HASH=$mode CUGRAPH_AMD_LOUVAIN_TRACE=1 timeout 600 python bench_louvain.py --scale 22 --cpu-scale 18 --out "$O/r3e_louvain_s22_hash$mode.json" 2>"$O/r3e_louvain_hash$mode.err" | cut -c1-420
  grep "\[louvain\]" "$O/r3e_louvain_hash$mode.err" | tail -8
done
LIBS="base cur p2fixlate base cur p2fixlate" BENCH_EXTRA="--no-extras" bash tools/gpu_ab.sh 2>&1 | tail -14
cp "$O/ab.log" "$O/r3e_ab.log"
CUGRAPH_AMD_BFS_TRACE=1 timeout 300 python bench_traversal.py --scale 24 --roots 2 --no-sssp --no-cpu-baseline --no-check 2>"$O/r3e_bfs_trace.err" | cut -c1-300
grep "\[bfs\]" "$O/r3e_bfs_trace.err" | tail -30
