#!/usr/bin/env bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29521 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
F='^\(HIP\|ROCm\|Hostname\|Librccl\|RCCL\|\[rank\|\[W\|/opt/amdgpu\)'
timeout 300 python -X faulthandler - <<PY 2>&1 | grep -v "$F" | tail -40 | cut -c1-300
import argparse, json, sys, os
sys.path.insert(0, "$R")
import torch
from cugraph_amd import mg
import cugraph_amd.mg as m
orig_ex = m.Exchange.__init__
def ex_init(self, *a, **k):
    print("stage: Exchange begin", flush=True); orig_ex(self, *a, **k); print("stage: Exchange done ncols", self.ncols, flush=True)
m.Exchange.__init__ = ex_init
orig_eng = m.HipLocalEngine.__init__
def eng_init(self, *a, **k):
    print("stage: engine begin", torch.cuda.memory_allocated() >> 20, "MiB", flush=True); orig_eng(self, *a, **k); print("stage: engine done", flush=True)
m.HipLocalEngine.__init__ = eng_init
orig_xe = m._exchange_edges
def xe(*a, **k):
    print("stage: exchange_edges begin", flush=True); r = orig_xe(*a, **k); print("stage: exchange_edges done", flush=True); return r
m._exchange_edges = xe
a = argparse.Namespace(scale=${SCALE:-26}, edge_factor=16, steps=20, warmup=3, hot_tile=None)
d = mg.bench_main(a)
print(json.dumps({k: d[k] for k in ("ms_per_step", "value", "graph_build_s", "exchange_rank0")}), d["roofline"])
PY
