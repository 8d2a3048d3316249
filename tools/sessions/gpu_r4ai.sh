#!/usr/bin/env bash
# round 4, last session: the full GPU suite, smoke and the default bench line on the committed tree
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee "$O/r4ai_pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$O/r4ai_smoke.log"
timeout 900 python bench.py 2>"$O/r4ai_bench.err" > "$O/r4ai_bench_s26.json"; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4ai_bench_s26.json").read().strip().splitlines()[-1])
r=d["roofline"]; print(d["value"], d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], "check", d["check"]["ok"], d["check"].get("converging_run", {}).get("iterations"),
  {k: (v.get("value"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("traffic") is not None, (v.get("check") or {}).get("ok")) for k, v in d["extra"].items()})
PY
