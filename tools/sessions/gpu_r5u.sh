#!/usr/bin/env bash
# GPU box, round 5 session u: the partitioned BFS with one host synchronisation less per level (the level's counters travel from the device)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_mg_capi.py tests/test_mg_traversal.py tests/test_reference_c_tests.py tests/test_pylibcugraph_on_gpu.py -m gpu -q -k "bfs or sssp or paths or traversal or int64" 2>&1 | tail -8 | tee "$O/r5u_tests.log"
for i in 1 2; do timeout 600 python bench_traversal.py --partitioned --transport ipc --scale 24 --weights int --roots 16 2>/dev/null | grep "^{" > "$O/r5u_part_ipc1_s24_$i.json"; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5u_part_ipc1_s24_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], {k: (d[k].get("mteps_harmonic_mean"), d[k].get("mean_ms"), (d[k].get("check") or {}).get("ok")) for k in ("bfs", "sssp") if k in d})
PY
