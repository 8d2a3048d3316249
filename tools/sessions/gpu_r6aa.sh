#!/usr/bin/env bash
# round 6: device-driven exchange of a partitioned top-down BFS level (counts + tuples from the device, no host): the MG traversal tests, then A/B
# against the host count matrix (CUGRAPH_AMD_MG_BFS_DEVICE_EXCHANGE=0) with 1 / 2 / 4 ranks on this one GPU
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${TAG:-r6aa}
timeout 900 python -m pytest tests/test_mg_capi.py -m gpu -q -x -k "bfs or windows or int64 or extract_paths or bad_argument" 2>&1 | tail -8 | tee "$O/${TAG}_pytest_mg_bfs.log"
{
for rep in 1 2; do
for dev in 1 0; do
  echo "== 1 rank, device exchange = $dev"
  CUGRAPH_AMD_MG_BFS_DEVICE_EXCHANGE=$dev timeout 600 python bench_traversal.py --partitioned --transport ipc --scale 24 --weights int --roots 16 --no-sssp --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['bfs']['ms_mean'], d['bfs']['ms_median'], d['bfs']['ms_min'], d['bfs']['rounds_mean'], d['bfs'].get('check',{}).get('ok'))"
done; done
for w in 2 4; do for dev in 1 0; do
  echo "== $w ranks sharing the GPU, device exchange = $dev"
  CUGRAPH_AMD_MG_BFS_DEVICE_EXCHANGE=$dev CUGRAPH_AMD_MG_TEST_SINGLE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench_traversal.py --gpus $w --transport ipc --scale 24 --weights int --roots 16 --no-sssp --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['bfs']['ms_mean'], d['bfs']['ms_median'], d['bfs']['ms_min'], d['bfs']['rounds_mean'], d['bfs'].get('check',{}).get('ok'))"
done; done
} 2>&1 | tee "$O/${TAG}_mg_bfs_device_exchange.txt"
