#!/bin/bash
# after the stand-in position fix of expand_frontier_mlp: traversal tests, the partitioned line, counter traffic on the final sources, the driver's line
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_mg_traversal.py tests/test_gpu_parity.py -x -q -m gpu -k "sssp or mg_" 2>&1 | tail -2 | tee $O/r3t_pytest.log
timeout 600 python bench_traversal.py --partitioned --scale 24 --weights int --roots 16 > $O/r3t_part.out 2> $O/r3t_part.err; echo "partitioned rc=$?"; grep "^{" $O/r3t_part.out > $O/r3z_partitioned_s24.json; cut -c1-900 $O/r3z_partitioned_s24.json
timeout 1500 python tools/traffic_collect.py > $O/r3t_traffic.log 2>&1; tail -3 $O/r3t_traffic.log | cut -c1-200
cp $O/traffic_latest.json profiles/traffic_latest.json
timeout 900 python bench.py 2>$O/r3t_bench.err > $O/r3t_bench_s26.json; echo "bench rc=$?"; cut -c1-200 $O/r3t_bench_s26.json
for sc in 22 24; do timeout 300 python bench.py --scale $sc --no-extras --cpu-scale 20 2>/dev/null > $O/r3t_bench_s$sc.json; cut -c1-120 $O/r3t_bench_s$sc.json; done
