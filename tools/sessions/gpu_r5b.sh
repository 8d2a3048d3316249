#!/usr/bin/env bash
# round 5, session b: phase 2 with the fma-based fixed point (4 VALU per partial instead of 32) and 1 / 2 batches in flight; overlapped
# iterations again on top of it (phase-1 workgroups 192..224, chunk order, CU masks); parity of everything PageRank (the row sums round
# differently now); SSSP / BFS with predecessors as the headline (parent sweep with 4 edges in flight).  Every step has its own timeout.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pagerank or overlapped or mggraph" --durations=5 2>&1 | tail -12 | tee "$O/r5b_tests.log"
timeout 300 python -m pytest tests/test_mg_capi.py -m gpu -x -q -k "pagerank" 2>&1 | tail -5 | tee -a "$O/r5b_tests.log"
OV=CUGRAPH_AMD_PR_OVERLAP
V26="base CUGRAPH_AMD_P2_BATCHES=2 $OV=192 $OV=200 $OV=208 $OV=216 $OV=224 $OV=208,CUGRAPH_AMD_P2_BATCHES=1 $OV=216,CUGRAPH_AMD_P2_BATCHES=1 $OV=208,CUGRAPH_AMD_PR_OVERLAP_ORDER=0 $OV=208,CUGRAPH_AMD_PR_OVERLAP_MASK=1 $OV=216,CUGRAPH_AMD_PR_OVERLAP_MASK=1"
timeout 150 python tools/plan_sweep.py --scale 26 --steps 20 --reps 2 $V26 2>&1 | grep "^rep" | tee "$O/r5b_sweep_s26.log"
S0=CUGRAPH_AMD_TP_STATIC_FRAC=0
V24="base $S0 $S0,$OV=200 $S0,$OV=208 $S0,$OV=216 $S0,$OV=224"
timeout 90 python tools/plan_sweep.py --scale 24 --steps 40 --reps 2 $V24 2>&1 | grep "^rep" | tee "$O/r5b_sweep_s24.log"
timeout 90 python tools/plan_sweep.py --scale 22 --steps 100 --reps 2 $V24 2>&1 | grep "^rep" | tee "$O/r5b_sweep_s22.log"
BEST=$(python - <<'PY'
import re, collections
best = collections.defaultdict(list)
for l in open("gpurun_out/r5b_sweep_s26.log"):
    m = re.match(r"rep \d+ (\S+)\s+ms/iter ([\d.]+)", l)
    if m and "OVERLAP" in m.group(1): best[m.group(1)].append(float(m.group(2)))
if best:
    k = min(best, key=lambda v: sum(best[v]) / len(best[v]))
    print(" ".join(kv for kv in k.split(",")))
PY
)
echo "best overlapped variant at RMAT-26: $BEST" | tee "$O/r5b_best.log"
P="$O/prof_r5b"; rm -rf "$P"; mkdir -p "$P"
( cd /tmp && export TMPDIR=/tmp && env $BEST timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$P/stats" -o run -- python "$R/bench.py" --scale 26 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$P/stats.log" 2>&1 )
tail -1 "$P/stats.log" | cut -c1-1200
python "$R/tools/rocpd_summary.py" "$P" 2>&1 | grep -i "kernel \|tiled_phase\|tiled_finish" | tee "$O/r5b_s26_rocprofv3_summary.txt"
python "$R/tools/rocpd_summary.py" --overlap "k_tiled_phase1" "k_tiled_phase2" "$P" 2>&1 | head -30 | tee -a "$O/r5b_s26_rocprofv3_summary.txt"
find "$P" -name "*.db" -delete
timeout 300 python bench_traversal.py --scale 24 --weights int --roots 16 --no-cpu-baseline --out "$O/r5b_traversal_s24_int.json" 2>&1 | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.read())
for k in ("bfs", "sssp"):
    x = d[k]; print(k, "with pred", x["mean_ms"], "ms frac", x["roofline"]["frac"], "| distance only", x.get("distance_only"), "| check", x.get("check", {}).get("ok"))'
