#!/usr/bin/env bash
# round 4, session ag: plumbing lines of the partitioned traversals (with the distributed fixed-point check) and of Louvain on several ranks sharing the GPU
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0 CUGRAPH_AMD_MG_TEST_SINGLE_GPU=1
for w in 2 4; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench_traversal.py --gpus $w --transport ipc --scale 24 --weights int --roots 8 2>"$O/r4ag_part_ipc${w}.err" | grep "^{" > "$O/r4ag_part_ipc${w}_s24.json"; echo "traversal ranks $w rc=$?"; tail -1 "$O/r4ag_part_ipc${w}.err" | cut -c1-200
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4ag_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], {k: ({kk: vv for kk, vv in v.items() if not isinstance(vv, (dict, list))} if isinstance(v, dict) else v) for k, v in d.items() if k in ("bfs", "sssp", "value", "n_gpus", "check", "modularity")})
PY
