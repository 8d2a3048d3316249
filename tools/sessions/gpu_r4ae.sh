#!/usr/bin/env bash
# round 4, session ae: the bench line with the converging run (epsilon 1e-6) in its check; PageRank at scales 23 and 25 (the curve of SURVEY section 8d)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
timeout 900 python bench.py 2>"$O/r4ae_bench.err" > "$O/r4ae_bench_s26.json"; echo "bench rc=$?"; tail -2 "$O/r4ae_bench.err" | cut -c1-300
for sc in 23 25; do timeout 400 python bench.py --scale $sc --no-extras --no-cpu-baseline 2>/dev/null > "$O/r4ae_bench_s$sc.json"; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4ae_bench_s*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "check", d["check"]["ok"], d["check"].get("converging_run"))
PY
