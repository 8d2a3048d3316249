#!/usr/bin/env bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
: > "$O/scales.log"
for scale in ${SCALES:-24 26}; do for kern in ${KERNS:-flat rows}; do for hot in ${HOTS:-16384}; do
  echo "== scale=$scale kernel=$kern hot=$hot" >> "$O/scales.log"
  CUGRAPH_AMD_PAGERANK_KERNEL=$kern timeout 400 python bench.py --scale $scale --steps 10 --warmup 2 --no-cpu-baseline --hot-tile $hot 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_kernel_ms'], d['roofline']['frac'], d['graph_build_s'])" >> "$O/scales.log" 2>&1
done; done; done
cat "$O/scales.log"
