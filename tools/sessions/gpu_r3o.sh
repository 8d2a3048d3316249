#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sssp and not subqueues" 2>&1 | tail -3 | tee $O/r3o_pytest.log
CUGRAPH_AMD_SSSP_TRACE=1 SWEEP=lh timeout 600 python tools/sssp_sweep.py 24 1 2>&1 | grep "^\[sssp\]" | tail -19 | head -9 | cut -c1-200
for u in 0 512 128 2048 0 512; do
  echo "== UB8 $u"; CUGRAPH_AMD_SSSP_UB8=$u SWEEP=lh timeout 600 python tools/sssp_sweep.py 24 16 2>&1 | grep mean | head -2 | cut -c1-250
done | tee $O/r3o_sssp_ub8.log
