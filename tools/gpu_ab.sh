#!/usr/bin/env bash
# GPU box: A/B of library variants in gpurun_libs/ (LIBS="a b"), bench.py at SCALE, results in gpurun_out/ab.log
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
cp cugraph_amd/lib/libcugraph_c.so /tmp/orig.so
: > "$O/ab.log"
fmt='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print(d["ms_per_step"], d["value"], r.get("avg_phase1_ms"), r.get("avg_phase2_ms"), r["frac"])'
for lib in ${LIBS:-base}; do for scale in ${SCALES:-26}; do
  cp "gpurun_libs/$lib.so" cugraph_amd/lib/libcugraph_c.so
  echo "== lib=$lib scale=$scale" >> "$O/ab.log"
  timeout 600 python bench.py --scale $scale --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline ${BENCH_EXTRA:-} 2>&1 | tail -3 | python -c "$fmt" >> "$O/ab.log" 2>&1
done; done
cp /tmp/orig.so cugraph_amd/lib/libcugraph_c.so
cat "$O/ab.log"
