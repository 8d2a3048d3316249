#!/usr/bin/env bash
# round 4, session aj: near / far windows in the partitioned SSSP: parity (all MG SSSP tests + schedule variants), A/B against one unbounded window at 1 / 2 / 4 ranks
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_mg_capi.py tests/test_reference_c_tests.py tests/test_c_conformance.py tests/test_pylibcugraph_on_gpu.py tests/test_mg_traversal.py -m gpu -x -q -k "sssp or SSSP or ranks or mg_sssp or mggraph" 2>&1 | tail -4 | tee "$O/r4aj_tests.log"
for win in 1 0; do
  CUGRAPH_AMD_MG_SSSP_WINDOW=$win timeout 600 python bench_traversal.py --partitioned --transport ipc --scale 24 --weights int --roots 16 2>/dev/null | grep "^{" > "$O/r4aj_part_ipc1_win$win.json"
  for w in 2 4; do
    CUGRAPH_AMD_MG_SSSP_WINDOW=$win CUGRAPH_AMD_MG_TEST_SINGLE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench_traversal.py --gpus $w --transport ipc --scale 24 --weights int --roots 8 2>/dev/null | grep "^{" > "$O/r4aj_part_ipc${w}_win$win.json"
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4aj_part_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    s=d["sssp"]; print(f.split("/")[-1], "sssp ms", s["ms_mean"], s["ms_median"], "rounds", s["rounds_mean"], "check", (s.get("check") or {}).get("ok"), "| bfs", d["bfs"]["ms_mean"])
PY
