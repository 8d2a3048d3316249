#!/usr/bin/env bash
# round 4, session ab: Louvain chunk kernel with (cluster, cluster weight) packed per vertex (one random access per edge instead of two dependent ones): parity, A/B
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_mg_capi.py -m gpu -x -q -k "louvain or Louvain" 2>&1 | tail -3 | tee "$O/r4ab_louvain_tests.log"
for sc in 22 26; do
  for pack in 1 0; do
    CUGRAPH_AMD_LOUVAIN_PACK=$pack timeout 900 python bench_louvain.py --scale $sc --cpu-scale 0 --out "$O/r4ab_louvain_s${sc}_pack$pack.json" > /dev/null 2>"$O/r4ab_louvain_s${sc}_pack$pack.err"; echo "s$sc pack=$pack rc=$?"
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4ab_louvain_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], "s", d.get("value"), d.get("seconds_all"), "Q", d.get("modularity"), "clusters", d.get("clusters"), "sweeps", d.get("sweeps"), "ok", (d.get("check") or {}).get("ok"))
PY
