#!/usr/bin/env bash
# round 5, session a: (1) the plain-C conformance binary 20x as the FIRST GPU processes of a fresh box (the one-off memory fault of round 4),
# (2) parity of the overlapped PageRank iterations, (3) sweep of the phase-1 workgroup count of the overlapped step at RMAT-26 / 24 / 22
# (one graph per scale, plans interleaved: tools/plan_sweep.py), (4) kernel trace of the best setting: do the two kernels run concurrently?
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
python - <<'PY' 2>&1 | tail -6 | tee "$O/r5a_conformance_loop.log"
import subprocess, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, "tests")
import test_c_conformance as t
exe = t.build_binary(Path(tempfile.mkdtemp()))
bad, t0 = 0, time.time()
for i in range(20):
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    if r.returncode != 0:
        bad += 1
        print("run", i, "rc", r.returncode, r.stdout[-600:])
print(f"conformance binary as the first GPU processes of the box: 20 runs, {bad} failed, {time.time() - t0:.1f} s")
PY
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "overlapped or is_reproducible or phase1_schedules or lds_tile_sizes" --durations=5 2>&1 | tail -12 | tee "$O/r5a_tests.log"
V26='base CUGRAPH_AMD_PR_OVERLAP=176 CUGRAPH_AMD_PR_OVERLAP=192 CUGRAPH_AMD_PR_OVERLAP=208 CUGRAPH_AMD_PR_OVERLAP=216 CUGRAPH_AMD_PR_OVERLAP=224 CUGRAPH_AMD_PR_OVERLAP=232 CUGRAPH_AMD_PR_OVERLAP=240 CUGRAPH_AMD_PR_OVERLAP=208,CUGRAPH_AMD_PR_OVERLAP_MASK=1 CUGRAPH_AMD_PR_OVERLAP=224,CUGRAPH_AMD_PR_OVERLAP_MASK=1'
timeout 600 python tools/plan_sweep.py --scale 26 --steps 20 --reps 2 $V26 2>&1 | grep "^rep" | tee "$O/r5a_sweep_s26.log"
S0=CUGRAPH_AMD_TP_STATIC_FRAC=0
V24="base $S0 $S0,CUGRAPH_AMD_PR_OVERLAP=208 $S0,CUGRAPH_AMD_PR_OVERLAP=224 $S0,CUGRAPH_AMD_PR_OVERLAP=232 $S0,CUGRAPH_AMD_PR_OVERLAP=240"
timeout 300 python tools/plan_sweep.py --scale 24 --steps 40 --reps 2 $V24 2>&1 | grep "^rep" | tee "$O/r5a_sweep_s24.log"
timeout 300 python tools/plan_sweep.py --scale 22 --steps 100 --reps 2 $V24 2>&1 | grep "^rep" | tee "$O/r5a_sweep_s22.log"
BEST=$(python - <<'PY'
import re, collections
best = collections.defaultdict(list)
for l in open("gpurun_out/r5a_sweep_s26.log"):
    m = re.match(r"rep \d+ (\S+)\s+ms/iter ([\d.]+)", l)
    if m and "OVERLAP" in m.group(1): best[m.group(1)].append(float(m.group(2)))
if best:
    k = min(best, key=lambda v: sum(best[v]) / len(best[v]))
    print(" ".join(kv for kv in k.split(",")))
PY
)
echo "best overlapped variant at RMAT-26: $BEST" | tee "$O/r5a_best.log"
P="$O/prof_r5a"; rm -rf "$P"; mkdir -p "$P"
cd /tmp && export TMPDIR=/tmp
env $BEST timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$P/stats" -o run -- python "$R/bench.py" --scale 26 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-check > "$P/stats.log" 2>&1
tail -2 "$P/stats.log" | cut -c1-1500
python "$R/tools/rocpd_summary.py" "$P" 2>&1 | head -12 | tee "$O/r5a_s26_rocprofv3_summary.txt"
python "$R/tools/rocpd_summary.py" --overlap "k_tiled_phase1" "k_tiled_phase2" "$P" 2>&1 | tee -a "$O/r5a_s26_rocprofv3_summary.txt"
find "$P" -name "*.db" -delete
