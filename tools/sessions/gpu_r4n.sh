#!/usr/bin/env bash
# round 4, session n: edge ids round trip; the partitioned traversals with ONE rank (library driver vs torch orchestration vs single-GPU entry point)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "edge_ids or decompress or create" 2>&1 | tail -4 | tee "$O/r4n_tests.log"
timeout 900 python bench_traversal.py --partitioned --transport ipc --scale 24 --weights int --roots 16 --out "$O/r4n_part_ipc1_s24.json" 2>"$O/r4n_ipc.err" | cut -c1-600
tail -3 "$O/r4n_ipc.err" | cut -c1-300
timeout 900 python bench_traversal.py --partitioned --transport rccl --scale 24 --weights int --roots 16 2>/dev/null | grep "^{" > "$O/r4n_part_rccl1_s24.json"; cut -c1-500 "$O/r4n_part_rccl1_s24.json"
timeout 600 python bench_traversal.py --scale 24 --weights int --roots 16 --no-cpu-baseline --out "$O/r4n_sg_s24.json" > /dev/null 2>&1
python - <<'PY'
import json
for f in ("r4n_part_ipc1_s24","r4n_part_rccl1_s24"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1]); print(f, "bfs", d["bfs"]["ms_median"], d["bfs"]["ms_mean"], "sssp", d.get("sssp",{}).get("ms_median"), d.get("sssp",{}).get("ms_mean"), "build", d["graph_build_s"])
    except Exception as e: print(f, "unreadable", e)
d=json.load(open("gpurun_out/r4n_sg_s24.json")); print("sg bfs", d["bfs"]["mean_ms"], "sssp", d["sssp"]["mean_ms"])
PY
