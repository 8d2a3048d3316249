#!/usr/bin/env bash
# round 4, session t: Louvain big-row kernel: threads per workgroup / table size variants (LIBS), same box
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
cp cugraph_amd/lib/libcugraph_c.so /tmp/orig.so
cp /tmp/orig.so gpurun_libs/head.so
for lib in head big1024 bigs4096 bigs2048 bigs4096t256; do
  cp "gpurun_libs/$lib.so" cugraph_amd/lib/libcugraph_c.so
  for sc in 22 26; do
    timeout 900 python bench_louvain.py --scale $sc --cpu-scale 0 --out "$O/r4t_louvain_s${sc}_$lib.json" > /dev/null 2>"$O/r4t_louvain_s${sc}_$lib.err"; echo "s$sc $lib rc=$?"
  done
done
cp /tmp/orig.so cugraph_amd/lib/libcugraph_c.so
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4t_louvain_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], "s", d.get("value"), d.get("seconds_all"), "Q", d.get("modularity"), "clusters", d.get("clusters"), "sweeps", d.get("sweeps"), "ok", (d.get("check") or {}).get("ok"))
PY
