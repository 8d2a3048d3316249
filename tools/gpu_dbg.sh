#!/bin/bash
# debugging session: which configuration of the one-rank multi-GPU PageRank loses / creates mass
O=gpurun_out; mkdir -p $O
run() { echo "=== $*" >> $O/debug_mg1d.log; env "$@" timeout 200 python tools/debug_mg1d.py 22 >> $O/debug_mg1d.log 2>&1; }
: > $O/debug_mg1d.log
run DBG_X=default
run CUGRAPH_AMD_POOL=0
run CUGRAPH_AMD_MG_OWN_STREAM=1
run DBG_DEVSYNC=1
grep -v "amdgpu.ids\|socket.cpp\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|Exception ignored\|Traceback\|pylib.py\|TypeError\|first differing" $O/debug_mg1d.log
