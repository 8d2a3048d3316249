#!/usr/bin/env bash
# GPU box, round 3 session B: clean phase-1 schedule sweep (static prefix, mixed order), bench line with extras, self-spawned 2-rank plumbing run
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
timeout 600 python tools/plan_sweep.py --scale 26 --steps 20 --reps 2 \
  "CUGRAPH_AMD_TP_STATIC_FRAC=0" \
  "CUGRAPH_AMD_TP_STATIC_FRAC=0.3" \
  "CUGRAPH_AMD_TP_STATIC_FRAC=0.5" \
  "CUGRAPH_AMD_TP_STATIC_FRAC=0.7" \
  "CUGRAPH_AMD_TP_STATIC_FRAC=0.5,CUGRAPH_AMD_TP_RUN_COST=2.0" \
  "CUGRAPH_AMD_TP_STATIC_FRAC=0,CUGRAPH_AMD_TP_ORDER=mix" \
  "CUGRAPH_AMD_TP_STATIC_FRAC=0,CUGRAPH_AMD_TP_CHUNK_BIG=64" \
  2>&1 | grep -v amdgpu.ids | tee "$O/r3b_sweep.log" | tail -16
timeout 900 python bench.py 2>"$O/r3b_bench.err" | tee "$O/r3b_bench_s26.json" | cut -c1-300
tail -5 "$O/r3b_bench.err"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3b_bench_s26.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], d["roofline"]["traffic_source"][:60] if d["roofline"]["traffic_source"] else None)
print("check", d.get("check"))
for k,v in d.get("extra",{}).items():
    if isinstance(v, dict):
        print(k, {x: v.get(x) for x in ("value","mean_ms","mean_levels","mean_steps","mean_relaxations_per_edge","sweeps")}, "frac", v.get("roofline",{}).get("frac"), "check", v.get("check"), "cpu", (v.get("cpu_baseline") or {}).get("value"))
    else:
        print(k, v)
PY
CUGRAPH_AMD_MG_TEST_SINGLE_GPU=1 timeout 600 python bench.py --gpus 2 --scale 22 --steps 5 --warmup 2 --cpu-scale 18 2>"$O/r3b_mg2.err" | tee "$O/r3b_mg2.json" | cut -c1-1500
tail -3 "$O/r3b_mg2.err"
