#!/usr/bin/env bash
# GPU box, round 3 session G: BFS on the symmetrised RMAT-26 graph (2^31 directed edges), Louvain at RMAT-26, the full bench line
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
timeout 900 python bench_traversal.py --scale 26 --symmetric --roots 8 --no-sssp --no-cpu-baseline --out "$O/r3g_traversal_s26_sym.json" 2>"$O/r3g_s26sym.err" | cut -c1-1400
tail -5 "$O/r3g_s26sym.err"
timeout 900 python bench_louvain.py --scale 26 --cpu-scale 0 --repeats 1 --out "$O/r3g_louvain_s26.json" 2>"$O/r3g_louvain26.err" | cut -c1-900
tail -3 "$O/r3g_louvain26.err"
/usr/bin/time -v timeout 900 python bench.py > "$O/r3g_bench_s26.json" 2> "$O/r3g_bench.err"; echo "bench rc=$?"
grep -E "Elapsed|Maximum resident|dumped|rror" "$O/r3g_bench.err" | head
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3g_bench_s26.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "l1-tracking", d["check"].get("ms_per_step_tracking_l1_change"), "check", d["check"]["ok"])
for k,v in d.get("extra",{}).items():
    if isinstance(v, dict):
        print(k, {x: v.get(x) for x in ("value","mean_ms","mean_steps","mean_relaxations_per_edge","sweeps")}, "frac", v.get("roofline",{}).get("frac"), "check", (v.get("check") or {}).get("ok"))
    else:
        print(k, v)
PY
