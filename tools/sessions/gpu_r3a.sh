#!/usr/bin/env bash
# GPU box, round 3 session A: phase-1 schedule variants (static chunk ranges, stretch mapping) -- parity subset, sweep, per-workgroup times, bench line
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "(pagerank and not config2 and not full_size) or release or goldens" 2>&1 | tail -8 | tee "$O/r3a_pytest.log"
timeout 600 python tools/plan_sweep.py --scale 26 --steps 20 --reps 2 \
  "CUGRAPH_AMD_TP_STATIC_FRAC=0,CUGRAPH_AMD_TP_STRETCH=0" \
  "CUGRAPH_AMD_TP_STATIC_FRAC=0,CUGRAPH_AMD_TP_STRETCH=1" \
  "CUGRAPH_AMD_TP_STATIC_FRAC=0.85,CUGRAPH_AMD_TP_STRETCH=0" \
  "CUGRAPH_AMD_TP_STATIC_FRAC=0.85,CUGRAPH_AMD_TP_STRETCH=1" \
  "CUGRAPH_AMD_TP_STATIC_FRAC=0.7,CUGRAPH_AMD_TP_STRETCH=1" \
  "CUGRAPH_AMD_TP_STATIC_FRAC=0.93,CUGRAPH_AMD_TP_STRETCH=1" \
  "CUGRAPH_AMD_TP_STATIC_FRAC=0.85,CUGRAPH_AMD_TP_STRETCH=1,CUGRAPH_AMD_TP_RUN_COST=0.8" \
  "CUGRAPH_AMD_TP_STATIC_FRAC=0.85,CUGRAPH_AMD_TP_STRETCH=1,CUGRAPH_AMD_TP_RUN_COST=2.0" \
  "CUGRAPH_AMD_TP_STATIC_FRAC=0.85,CUGRAPH_AMD_TP_STRETCH=1,CUGRAPH_AMD_TP_TAIL_CHUNK=4" \
  "CUGRAPH_AMD_TP_STATIC_FRAC=0.85,CUGRAPH_AMD_TP_STRETCH=1,CUGRAPH_AMD_TP_TAIL_CHUNK=16" \
  2>&1 | tee "$O/r3a_sweep.log" | tail -24
CUGRAPH_AMD_TILED_DEBUG=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check 2>&1 | grep -E "tiled|\{" | cut -c1-600 | tee "$O/r3a_dbg.log"
CUGRAPH_AMD_BUILD_TRACE=1 timeout 600 python bench.py 2>"$O/r3a_bench.err" | tee "$O/r3a_bench_s26.json" | cut -c1-900
grep "\[build\]" "$O/r3a_bench.err" | tail -40
