#!/usr/bin/env bash
# GPU box: tests + PageRank kernel/hot-tile sweep at one scale.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests -m gpu -x -q > "$O/pytest_gpu.log" 2>&1; echo "pytest_rc=$?" >> "$O/pytest_gpu.log"
SCALE="${SCALE:-22}"
: > "$O/sweep.log"
for kern in flat rows; do for hot in ${HOTS:-0 8192 16384}; do
  echo "== kernel=$kern hot=$hot" >> "$O/sweep.log"
  CUGRAPH_AMD_PAGERANK_KERNEL=$kern timeout 120 python bench.py --scale "$SCALE" --steps 20 --warmup 3 --no-cpu-baseline --hot-tile $hot 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['avg_kernel_ms'], d['roofline']['frac'], d['graph_build_s'])" >> "$O/sweep.log"
done; done
tail -4 "$O/pytest_gpu.log"; cat "$O/sweep.log"
