#!/usr/bin/env bash
# round 4, session v: Louvain chunk kernel with gains per table slot (new) against gains per edge (prev), same box; Louvain tests on new
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_mg_capi.py tests/test_c_conformance.py tests/test_pylibcugraph_on_gpu.py -m gpu -x -q -k "louvain or Louvain" 2>&1 | tail -3 | tee "$O/r4v_louvain_tests.log"
cp cugraph_amd/lib/libcugraph_c.so /tmp/orig.so
for lib in ${LIBS:-prev new}; do
  cp "gpurun_libs/$lib.so" cugraph_amd/lib/libcugraph_c.so
  for sc in 22 26; do
    timeout 900 python bench_louvain.py --scale $sc --cpu-scale 0 --out "$O/${TAG:-r4v}_louvain_s${sc}_$lib.json" > /dev/null 2>"$O/${TAG:-r4v}_louvain_s${sc}_$lib.err"; echo "s$sc $lib rc=$?"
  done
done
cp /tmp/orig.so cugraph_amd/lib/libcugraph_c.so
python - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/%s_louvain_*.json" % os.environ.get("TAG","r4v"))):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], "s", d.get("value"), d.get("seconds_all"), "Q", d.get("modularity"), "clusters", d.get("clusters"), "sweeps", d.get("sweeps"), "ok", (d.get("check") or {}).get("ok"))
PY
