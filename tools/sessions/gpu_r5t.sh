#!/usr/bin/env bash
# GPU box, round 5 session t: the rest of the scale curve on the final sources (22 / 24 / 26 are in the r5z session): RMAT-23, 25, 27
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
for sc in 23 25; do timeout 300 python bench.py --scale $sc --no-extras --cpu-scale 20 2>/dev/null > "$O/r5z_bench_s$sc.json"; cut -c1-150 "$O/r5z_bench_s$sc.json"; done
timeout 500 python bench.py --scale 27 --no-extras --no-cpu-baseline 2>"$O/r5z_bench_s27.err" > "$O/r5z_bench_s27.json"; cut -c1-150 "$O/r5z_bench_s27.json"; tail -2 "$O/r5z_bench_s27.err"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5z_bench_s2[357].json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f.split("/")[-1], d["ms_per_step"], "frac", r["frac"], "kernels", r.get("frac_kernels"), "check", (d.get("check") or {}).get("ok"))
    except Exception as e:
        print(f, "ERR", e)
PY
