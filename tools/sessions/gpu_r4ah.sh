#!/usr/bin/env bash
# round 4, session ah: bench.py --gpus N (ranks sharing the GPU) with the assembled-vector check against the single-GPU entry point
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
for w in 2 4 8; do CUGRAPH_AMD_MG_TEST_SINGLE_GPU=1 timeout 600 python bench.py --gpus $w --scale 24 --steps 10 --warmup 2 --no-cpu-baseline 2>"$O/r4ah_ipc${w}.err" > "$O/r4ah_ipc${w}_s24.json"; echo "ranks $w rc=$?"; tail -1 "$O/r4ah_ipc${w}.err" | cut -c1-200; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4ah_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    c=d["check"]; print(f.split("/")[-1], d["ms_per_step"], "ok", c["ok"], "mass", c["mass_err"], {k: v for k, v in c.get("vs_single_gpu", {}).items() if k != "what"})
PY
