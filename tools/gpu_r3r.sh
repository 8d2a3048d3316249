#!/bin/bash
# BFS with EX_U edges in flight per lane: parity tests, then A/B against the one-edge walk (same box, interleaved)
O=gpurun_out; mkdir -p $O; cp cugraph_amd/lib/libcugraph_c.so /tmp/orig.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_c_conformance.py -x -q -m gpu -k "bfs or sssp or conformance or traversal" 2>&1 | tail -3 | tee $O/r3r_pytest.log
: > $O/r3r_bfs_ab.log
for lib in bfsnomlp head bfsnomlp head; do
  cp gpurun_libs/$lib.so cugraph_amd/lib/libcugraph_c.so
  for sym in "" "--symmetric"; do
    timeout 300 python bench_traversal.py --scale 24 --weights int --roots 32 --no-cpu-baseline --no-check --no-sssp $sym 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['bfs']; print('$lib', '$sym', 'bfs mean_ms', b['mean_ms'], 'min', b.get('min_ms'), 'max', b.get('max_ms'), 'MTEPS', d['value'])" >> $O/r3r_bfs_ab.log
  done
done
cp /tmp/orig.so cugraph_amd/lib/libcugraph_c.so
cat $O/r3r_bfs_ab.log
