#!/usr/bin/env bash
# round 4, session af: source-tile size (x entries staged in LDS per workgroup) at the small scales: does a smaller tile pay at RMAT-22 / 23 / 24?
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
for sc in 22 23 24; do for t in 0 8192 12288 16384 24576; do
  if [ $t -eq 0 ]; then extra=""; else extra="--hot-tile $t"; fi
  timeout 300 python bench.py --scale $sc --no-extras --no-cpu-baseline --no-check $extra 2>/dev/null > "$O/r4af_s${sc}_t$t.json"
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4af_s*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    r=d["roofline"]; print(f.split("/")[-1], d["ms_per_step"], "frac", r["frac"], "p1/p2", r["avg_phase1_ms"], r["avg_phase2_ms"], "plan_s", d["plan_build_s"])
PY
