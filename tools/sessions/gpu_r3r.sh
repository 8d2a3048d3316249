#!/bin/bash
# BFS pre-test of the cumulative visited word: agent-scope load (head) against the non-temporal load (bfsnt); same box, interleaved
O=gpurun_out; mkdir -p $O; cp cugraph_amd/lib/libcugraph_c.so /tmp/orig.so
: > $O/r3r_bfs_ab.log
for lib in bfsnt head bfsnt head; do
  cp gpurun_libs/$lib.so cugraph_amd/lib/libcugraph_c.so
  for mode in "" "topdown"; do
    CUGRAPH_AMD_BFS=$mode timeout 300 python bench_traversal.py --scale 24 --weights int --roots 32 --no-cpu-baseline --no-check --no-sssp 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['bfs']; print('$lib', 'mode=$mode', 'bfs mean_ms', b['mean_ms'], 'min', b.get('min_ms'), 'max', b.get('max_ms'), 'MTEPS', d['value'])" >> $O/r3r_bfs_ab.log
  done
done
cp /tmp/orig.so cugraph_amd/lib/libcugraph_c.so
cat $O/r3r_bfs_ab.log
