#!/usr/bin/env bash
# round 4, session u: Louvain: threads of the mid-row kernel (LIBS), kernel trace of the RMAT-26 call on the new defaults, Louvain tests on the new defaults
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_mg_capi.py -m gpu -x -q -k "louvain or Louvain" 2>&1 | tail -3 | tee "$O/r4u_louvain_tests.log"
cp cugraph_amd/lib/libcugraph_c.so /tmp/orig.so
cp /tmp/orig.so gpurun_libs/head.so
for lib in head midt1024 midt256; do
  cp "gpurun_libs/$lib.so" cugraph_amd/lib/libcugraph_c.so
  for sc in 22 26; do
    timeout 900 python bench_louvain.py --scale $sc --cpu-scale 0 --out "$O/r4u_louvain_s${sc}_$lib.json" > /dev/null 2>"$O/r4u_louvain_s${sc}_$lib.err"; echo "s$sc $lib rc=$?"
  done
done
cp /tmp/orig.so cugraph_amd/lib/libcugraph_c.so
P="$O/prof_r4u_louvain26"; rm -rf "$P"; mkdir -p "$P"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats -d "$P/stats" -o run -- python "$R/bench_louvain.py" --scale 26 --cpu-scale 0 --repeats 1 > "$P/stats.log" 2>&1 )
python tools/rocpd_summary.py "$P" > "$P/summary.txt" 2>&1
find "$P" -name "*.db" -delete
head -28 "$P/summary.txt" | cut -c1-150
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4u_louvain_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], "s", d.get("value"), d.get("seconds_all"), "Q", d.get("modularity"), "clusters", d.get("clusters"), "sweeps", d.get("sweeps"), "ok", (d.get("check") or {}).get("ok"))
PY
