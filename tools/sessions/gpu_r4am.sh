#!/usr/bin/env bash
# round 4, session am: direction-switch constants of the single-GPU BFS at RMAT-24 (environment only; nothing changes in the tree)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
for ab in "60 24" "30 24" "120 24" "240 24" "60 12" "60 48"; do set -- $ab
  CUGRAPH_AMD_BFS_ALPHA=$1 CUGRAPH_AMD_BFS_BETA=$2 timeout 300 python bench_traversal.py --scale 24 --weights int --no-sssp --no-cpu-baseline --roots 32 --out "$O/r4am_bfs_a$1_b$2.json" > /dev/null 2>&1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4am_bfs_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); b=d["bfs"]
    print(f.split("/")[-1], "mean ms", b["mean_ms"], "min", b["min_ms"], "max", b["max_ms"], "GTEPS", round(b["harmonic_mean_mteps"]/1e3,1), "levels", b["mean_levels"], "check", (b.get("check") or {}).get("ok"))
PY
