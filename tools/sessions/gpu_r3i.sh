#!/usr/bin/env bash
# GPU box, round 3 session I: Louvain at RMAT-26 (kernel stats, hash vs sorted), full bench line, RMAT-22 schedule sweep
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/prof_r3i_louvain26"; CUGRAPH_AMD_LOUVAIN_TRACE=1 timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof_r3i_louvain26" -o run -- python "$R/bench_louvain.py" --scale 26 --cpu-scale 0 --repeats 1 > "$O/r3i_louvain26_prof.log" 2>&1
python "$R/tools/rocpd_summary.py" "$O/prof_r3i_louvain26" > "$O/r3i_louvain_s26_rocprofv3_summary.txt" 2>&1; find "$O/prof_r3i_louvain26" -name "*.db" -delete
grep "\[louvain\]" "$O/r3i_louvain26_prof.log" | tail -5
head -22 "$O/r3i_louvain_s26_rocprofv3_summary.txt" | cut -c1-150
cd "$R"
CUGRAPH_AMD_LOUVAIN_HASH=0 CUGRAPH_AMD_LOUVAIN_TRACE=1 timeout 600 python bench_louvain.py --scale 26 --cpu-scale 0 --repeats 1 2>&1 | grep -E "\[louvain\]|value" | tail -5 | cut -c1-300
timeout 900 python bench.py > "$O/r3i_bench_s26.json" 2> "$O/r3i_bench.err"; echo "bench rc=$?"
grep -E "dumped|rror" "$O/r3i_bench.err" | head -5
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3i_bench_s26.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "l1-tracking", d["check"].get("ms_per_step_tracking_l1_change"), "check", d["check"]["ok"])
for k,v in d.get("extra",{}).items():
    if isinstance(v, dict):
        print(k, {x: v.get(x) for x in ("value","mean_ms","mean_steps","mean_relaxations_per_edge","sweeps")}, "frac", v.get("roofline",{}).get("frac"), "check", (v.get("check") or {}).get("ok"))
    else:
        print(k, v)
PY
bash tools/gpu_r3h.sh
