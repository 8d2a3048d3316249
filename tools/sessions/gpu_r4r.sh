#!/usr/bin/env bash
# round 4, session r: Louvain, rows of more than 4096 edges in LDS tables too (several work items per row): parity, A/B against CUGRAPH_AMD_LOUVAIN_BIG=0, kernel trace
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_mg_capi.py tests/test_c_conformance.py tests/test_pylibcugraph_on_gpu.py -m gpu -x -q -k "louvain or Louvain" 2>&1 | tail -5 | tee "$O/r4r_louvain_tests.log"
for sc in 22 26; do
  for mid in 1 0; do export CUGRAPH_AMD_LOUVAIN_BIG=$mid;
    timeout 900 python bench_louvain.py --scale $sc --cpu-scale 0 --out "$O/r4r_louvain_s${sc}_mid$mid.json" > /dev/null 2>"$O/r4r_louvain_s${sc}_mid$mid.err"; echo "s$sc mid=$mid rc=$?"
  done
done
unset CUGRAPH_AMD_LOUVAIN_BIG
P="$O/prof_r4r_louvain"; rm -rf "$P"; mkdir -p "$P"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$P/stats" -o run -- python "$R/bench_louvain.py" --scale 22 --cpu-scale 0 --repeats 3 > "$P/stats.log" 2>&1 )
python tools/rocpd_summary.py "$P" > "$P/summary.txt" 2>&1
find "$P" -name "*.db" -delete
head -40 "$P/summary.txt" | cut -c1-150
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4r_louvain_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], "s", d.get("value"), d.get("seconds_all"), "Q", d.get("modularity"), "clusters", d.get("clusters"), "sweeps", d.get("sweeps"), "frac", d["roofline"]["frac"], "check", d.get("check"))
PY
