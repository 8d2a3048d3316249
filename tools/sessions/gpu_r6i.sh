#!/usr/bin/env bash
# round 6, session i: SSSP two-stage wide rounds (filter survivors compacted per wavefront): parity, then A/B in one session: two-stage / one-stage filtered / no filter
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sssp" 2>&1 | tail -5 | tee "$O/r6i_tests.log"
fmt='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        j=json.loads(l); s=j["sssp"]; print("sssp mean_ms", s["mean_ms"], "frac", s["roofline"]["frac"], "dist-only", (s.get("distance_only") or {}).get("mean_ms"), "check", (j.get("check") or {}).get("ok"))'
for rep in 1 2; do for cfg in ${CFGS:-"F=1,S=1" "F=1,S=0" "F=0,S=0"}; do for w in ${WEIGHTS:-int unit}; do
  H=$(echo $cfg | grep -o "H=[0-9]*" | cut -d= -f2); W=$(echo $cfg | grep -o "W=[0-9.]*" | cut -d= -f2); O=$(echo $cfg | grep -o "O=[0-9]*" | cut -d= -f2); F=$(echo $cfg | sed 's/.*F=\([0-9]\).*/\1/'); S=$(echo $cfg | sed 's/.*S=\([0-9]\).*/\1/'); G=$(echo $cfg | grep -o "G=[0-9]*" | cut -d= -f2)
  echo -n "rep $rep filter=$F two_stage=$S grid=${G:-4} order=${O:-0} sweep=${W:-0.25} hot=${H:-1} weights=$w: "
  CUGRAPH_AMD_SSSP_HOT=${H:-1} CUGRAPH_AMD_SSSP_SWEEP=${W:-0.25} CUGRAPH_AMD_SSSP_ORDER=${O:-0} CUGRAPH_AMD_SSSP_FILTER=$F CUGRAPH_AMD_SSSP_TWO_STAGE=$S CUGRAPH_AMD_SSSP_S2_GRID=${G:-4} timeout 300 python bench_traversal.py --scale 24 --roots ${ROOTS:-16} --weights $w --no-cpu-baseline 2>/dev/null | python -c "$fmt"
done; done; done | tee "$O/${TAG:-r6i}_sssp_ab.txt"
CUGRAPH_AMD_SSSP_TRACE=1 timeout 300 python bench_traversal.py --scale 24 --roots 1 --weights int --no-cpu-baseline --no-check --single-variant 2>&1 | grep "^\[sssp\]" | tail -20 | cut -c1-250 > "$O/${TAG:-r6i}_sssp_trace.txt"
head -12 "$O/${TAG:-r6i}_sssp_trace.txt"
