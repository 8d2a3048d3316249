#!/bin/bash
# partitioned BFS with bottom-up levels: tests, then the one-rank RCCL measurement at RMAT-24 (heuristic / top-down only)
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_mg_traversal.py -x -q -m gpu 2>&1 | tail -4 | tee $O/r3s_pytest.log
for mode in "" topdown; do
  CUGRAPH_AMD_MG_BFS=$mode timeout 600 python bench_traversal.py --partitioned --scale 24 --weights int --roots 16 --no-sssp 2>$O/r3s_part.err | grep "^{" > $O/r3s_partitioned_s24_bfs_${mode:-auto}.json
  echo "direction=${mode:-auto}"; cut -c1-700 $O/r3s_partitioned_s24_bfs_${mode:-auto}.json
done
