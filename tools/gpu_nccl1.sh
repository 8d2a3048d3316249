#!/usr/bin/env bash
# GPU box: the multi-GPU code paths with a ONE-rank RCCL process group (the collectives run through RCCL, not gloo)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
F='^\(HIP\|ROCm\|Hostname\|Librccl\|RCCL\|\[rank\|\[W\|/opt/amdgpu\)'
for own in 0 1; do
echo "== pagerank mg, 1 rank nccl, scale ${SCALE:-22}, CUGRAPH_AMD_MG_OWN_STREAM=$own"
MASTER_PORT=$((29511 + own)) CUGRAPH_AMD_MG_OWN_STREAM=$own timeout 300 python - <<PY 2>&1 | grep -v "$F" | tail -4 | cut -c1-400
import argparse, json, sys
sys.path.insert(0, "$R")
from cugraph_amd import mg
a = argparse.Namespace(scale=${SCALE:-22}, edge_factor=16, steps=${STEPS:-50}, warmup=5, hot_tile=None)
d = mg.bench_main(a)
print(json.dumps({k: d[k] for k in ("ms_per_step", "value", "graph_build_s")}), d["roofline"]["avg_kernel_ms"])
PY
done
if [ "${TRAV:-1}" = 1 ]; then
echo "== traversal partitioned, 1 rank nccl, scale ${SCALE:-22}"
MASTER_PORT=29515 timeout 300 python bench_traversal.py --scale ${SCALE:-22} --roots 8 --weights int --partitioned --predecessors 2>&1 | grep -v "$F" | tail -3
fi
