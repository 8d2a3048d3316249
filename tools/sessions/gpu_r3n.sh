#!/bin/bash
# round 3, session N: the stream-sharing fix of the multi-GPU engines (one-rank RCCL runs)
R=$(pwd); O=gpurun_out; mkdir -p $O; TAG=r3n
timeout 900 python -m pytest tests/test_mg.py -x -q -m gpu 2>&1 | tail -4 | tee $O/${TAG}_pytest_mg.log
# the late-collective test must FAIL on the old behaviour (the engine "borrowing" torch's null stream)
cat > /tmp/old_stream.py <<'PY'
import sys, runpy
sys.argv = sys.argv[1:]
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from cugraph_amd import mg
def old(handle, dev):
    handle.set_stream(torch.cuda.current_stream().cuda_stream)   # 0: the library keeps its own stream
    return None
mg._share_stream = old
runpy.run_path("tests/mg_worker.py", run_name="__main__")
PY
mkdir -p /tmp/old1 /tmp/new1
for m in hip_nccl hip2d_nccl; do
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 python /tmp/old_stream.py tests/mg_worker.py $m 14 /tmp/old1 0.0 12 > /tmp/old1/log 2>&1; cp /tmp/old1/rank0.npz /tmp/old1/$m.npz
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29562 python tests/mg_worker.py $m 14 /tmp/new1 0.0 12 > /tmp/new1/log 2>&1; cp /tmp/new1/rank0.npz /tmp/new1/$m.npz
  python - $m <<'PY' | tee -a $O/r3n_old_vs_new.log
import sys, numpy as np
m = sys.argv[1]
a, b = np.load(f"/tmp/old1/{m}.npz"), np.load(f"/tmp/new1/{m}.npz")
x, y = np.zeros(1 << 14), np.zeros(1 << 14)
x[a["v"]] = a["x"]; y[b["v"]] = b["x"]
print(m, "old behaviour: mass", x.sum(), " fixed: mass", y.sum(), " max |old - fixed|", np.abs(x - y).max())
PY
done
DBG_LAYOUTS=1d,2d,1d,2d,1d,2d timeout 400 python tools/debug_mg1d.py 22 24 2>&1 | grep "mass\|max" | tee $O/${TAG}_mass.log
for lay in 1d 2d; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --layout $lay --scale 24 --steps 10 --warmup 2 --cpu-scale 18 2>"$O/${TAG}_nccl1_$lay.err" > "$O/${TAG}_nccl1_$lay.json"; echo "nccl world-1 layout $lay rc=$?"
  python - "$O/${TAG}_nccl1_$lay.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["layout"], d["value"], d["ms_per_step"], d["check"], d["phase_split_ms"], d["roofline"]["frac"])
PY
done
