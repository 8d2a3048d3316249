#!/usr/bin/env bash
# round 4, session s: Louvain big rows with staged (cluster, fixed-point weight) per edge; A/B BIG=1/0; k_lv_hash_chunks with 512 / 1024 threads per workgroup
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_mg_capi.py -m gpu -x -q -k "louvain or Louvain" 2>&1 | tail -5 | tee "$O/r4s_louvain_tests.log"
for sc in 22 26; do
  for big in 1 0; do
    CUGRAPH_AMD_LOUVAIN_BIG=$big timeout 900 python bench_louvain.py --scale $sc --cpu-scale 0 --out "$O/r4s_louvain_s${sc}_big$big.json" > /dev/null 2>"$O/r4s_louvain_s${sc}_big$big.err"; echo "s$sc big=$big rc=$?"
  done
done
cp cugraph_amd/lib/libcugraph_c.so /tmp/orig.so
for lib in lvh512 lvh1024; do
  cp "gpurun_libs/$lib.so" cugraph_amd/lib/libcugraph_c.so
  for sc in 22 26; do
    timeout 900 python bench_louvain.py --scale $sc --cpu-scale 0 --out "$O/r4s_louvain_s${sc}_$lib.json" > /dev/null 2>"$O/r4s_louvain_s${sc}_$lib.err"; echo "s$sc $lib rc=$?"
  done
done
cp /tmp/orig.so cugraph_amd/lib/libcugraph_c.so
P="$O/prof_r4s_louvain"; rm -rf "$P"; mkdir -p "$P"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$P/stats" -o run -- python "$R/bench_louvain.py" --scale 22 --cpu-scale 0 --repeats 3 > "$P/stats.log" 2>&1 )
python tools/rocpd_summary.py "$P" > "$P/summary.txt" 2>&1
find "$P" -name "*.db" -delete
head -24 "$P/summary.txt" | cut -c1-150
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4s_louvain_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], "s", d.get("value"), d.get("seconds_all"), "Q", d.get("modularity"), "clusters", d.get("clusters"), "sweeps", d.get("sweeps"), "frac", d["roofline"]["frac"], "ok", (d.get("check") or {}).get("ok"))
PY
