#!/bin/bash
# partitioned SSSP with owned destinations relaxed in place: tests, then one rank over RCCL at RMAT-24 with and without
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_mg_traversal.py -x -q -m gpu -k "sssp" 2>&1 | tail -3 | tee $O/r3v_pytest.log
for ip in 1 0; do
  CUGRAPH_AMD_MG_SSSP_INPLACE=$ip timeout 500 python bench_traversal.py --partitioned --scale 24 --weights int --roots 12 2>/dev/null | grep "^{" > $O/r3v_partitioned_s24_inplace$ip.json
  python - $O/r3v_partitioned_s24_inplace$ip.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read()); print(sys.argv[1][-14:], "sssp median", d["sssp"]["ms_median"], "mean", d["sssp"]["ms_mean"], "rounds", d["sssp"]["rounds_mean"], "| bfs median", d["bfs"]["ms_median"])
PY
done
