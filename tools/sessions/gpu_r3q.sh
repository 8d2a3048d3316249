#!/bin/bash
# A/B of library variants on the SSSP sweep (one process per variant, same box)
O=gpurun_out; mkdir -p $O; cp cugraph_amd/lib/libcugraph_c.so /tmp/orig.so
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sssp and not subqueues" 2>&1 | tail -2
: > $O/r3q_sssp_variants.log
for lib in ${LIBS:-nomlp u2 u4 u8 nomlp u4}; do
  cp gpurun_libs/$lib.so cugraph_amd/lib/libcugraph_c.so
  echo "== $lib" >> $O/r3q_sssp_variants.log
  SWEEP=lh timeout 300 python tools/sssp_sweep.py 24 16 2>&1 | grep mean | cut -c1-250 >> $O/r3q_sssp_variants.log
done
cp /tmp/orig.so cugraph_amd/lib/libcugraph_c.so
cat $O/r3q_sssp_variants.log
