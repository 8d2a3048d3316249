#!/usr/bin/env bash
# GPU box: the multi-GPU code paths with a ONE-rank RCCL process group (the collectives run through RCCL, not gloo)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
echo "== pagerank mg, 1 rank nccl, scale ${SCALE:-22}"
timeout 300 python - <<PY 2>&1 | grep -v '^\(HIP\|ROCm\|Hostname\|Librccl\|\[rank\)' | tail -8
import argparse, json, sys
sys.path.insert(0, "$R")
from cugraph_amd import mg
a = argparse.Namespace(scale=${SCALE:-22}, edge_factor=16, steps=10, warmup=2, hot_tile=None)
print(json.dumps(mg.bench_main(a)))
PY
echo "== traversal partitioned, 1 rank nccl, scale ${SCALE:-22}"
MASTER_PORT=29512 timeout 300 python bench_traversal.py --scale ${SCALE:-22} --roots 8 --weights int --partitioned --predecessors 2>&1 | grep -v '^\(HIP\|ROCm\|Hostname\|Librccl\|\[rank\)' | tail -8
echo "== traversal single-GPU path, scale ${SCALE:-22}"
timeout 300 python bench_traversal.py --scale ${SCALE:-22} --roots 8 --weights int --predecessors 2>&1 | tail -2
