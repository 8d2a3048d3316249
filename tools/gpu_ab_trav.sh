#!/usr/bin/env bash
# GPU box: A/B of library variants in gpurun_libs/ (LIBS="a b") on the BFS level trace of bench_traversal.py (bottom-up level times)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
cp cugraph_amd/lib/libcugraph_c.so /tmp/orig.so
for lib in ${LIBS:-bu_base}; do for envs in ${ENVS:-X=0}; do
  cp "gpurun_libs/$lib.so" cugraph_amd/lib/libcugraph_c.so
  echo "== lib=$lib env=$envs"
  env $envs CUGRAPH_AMD_BFS_TRACE=1 timeout 300 python bench_traversal.py --scale 24 --roots ${ROOTS:-8} --weights ${WEIGHTS:-unit} ${SSSP:---no-sssp} --no-cpu-baseline 2>"$O/ab_trav.err" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); d=j['bfs']; print('bfs mean_ms', d['mean_ms'], 'min', d['min_ms'], 'max', d['max_ms'], 'sssp', (j.get('sssp') or {}).get('mean_ms'))"
  python - "$O/ab_trav.err" <<'PY'
import sys,re
prev=None; acc={}
for l in open(sys.argv[1]):
    m=re.match(r"\[bfs\]\s+([0-9.]+) us\s+(\w+) (\d+) (\d+)", l)
    if not m: continue
    t=float(m.group(1)); kind=m.group(2)
    if kind=="init": prev=t; continue
    if prev is not None and kind=="bottom_up": acc.setdefault(kind,[]).append(t-prev)
    prev=t
b=acc.get("bottom_up",[])
print("bottom-up levels: n=%d mean %.0f us, sorted:"%(len(b), sum(b)/max(len(b),1)), " ".join("%.0f"%x for x in sorted(b)[-12:]))
PY
done; done
cp /tmp/orig.so cugraph_amd/lib/libcugraph_c.so
