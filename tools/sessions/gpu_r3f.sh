#!/usr/bin/env bash
# GPU box, round 3 session F: full -m gpu suite, Louvain kernel profile, RMAT-22 bench line, 2-D layout plumbing run
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee "$O/r3f_pytest.log"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/prof_r3f_louvain"; timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_r3f_louvain" -o run -- python "$R/bench_louvain.py" --scale 22 --cpu-scale 0 --repeats 2 > "$O/r3f_louvain_prof.log" 2>&1
python "$R/tools/rocpd_summary.py" "$O/prof_r3f_louvain" > "$O/r3f_louvain_s22_rocprofv3_summary.txt" 2>&1; find "$O/prof_r3f_louvain" -name "*.db" -delete
head -32 "$O/r3f_louvain_s22_rocprofv3_summary.txt" | cut -c1-150
cd "$R"
timeout 300 python bench.py --scale 22 --no-extras --cpu-scale 20 2>/dev/null | tee "$O/r3f_bench_s22.json" | cut -c1-250
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3f_bench_s22.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("s22 ms/step", d["ms_per_step"], "frac", r["frac"], "p1", r["avg_phase1_ms"], "p2", r["avg_phase2_ms"], "check", d["check"]["ok"])
PY
CUGRAPH_AMD_MG_TEST_SINGLE_GPU=1 timeout 600 python bench.py --gpus 4 --layout 2d --scale 22 --steps 5 --warmup 2 --no-cpu-baseline 2>"$O/r3f_mg2d.err" | tee "$O/r3f_mg2d.json" | cut -c1-1800
tail -3 "$O/r3f_mg2d.err"
