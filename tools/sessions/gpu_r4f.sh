#!/usr/bin/env bash
# round 4, session f: MG PageRank (fused small kernels, live const rows, one-rank direct mode): tests, then the one-rank lines against the SG entry point
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_mg_capi.py -m gpu -x -q 2>&1 | tail -8 | tee "$O/r4f_mgcapi.log"
one() { timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --gpus 2 "$@"; }
for sc in 26 24; do
  one --scale $sc --steps 20 --warmup 3 --no-cpu-baseline 2>"$O/r4f_ipc1_s$sc.err" > "$O/r4f_ipc1_s$sc.json"; echo "ipc one rank s$sc rc=$?"; tail -1 "$O/r4f_ipc1_s$sc.err" | cut -c1-300
  CUGRAPH_AMD_MG_PUSH_SELF=1 one --scale $sc --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null > "$O/r4f_ipc1_s${sc}_pushself.json"
  timeout 600 python bench.py --scale $sc --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-check 2>/dev/null > "$O/r4f_sg_s$sc.json"
done
for w in 2 4 8; do
  CUGRAPH_AMD_MG_TEST_SINGLE_GPU=1 timeout 600 python bench.py --gpus $w --scale 24 --steps 10 --warmup 2 --no-cpu-baseline 2>"$O/r4f_ipc${w}_s24.err" > "$O/r4f_ipc${w}_s24.json"; echo "ipc $w ranks on one GPU rc=$?"; tail -1 "$O/r4f_ipc${w}_s24.err" | cut -c1-300
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4f_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], "ms/step", d.get("ms_per_step"), "frac", (d.get("roofline") or {}).get("frac"), "p1/p2", (d.get("roofline") or {}).get("avg_phase1_ms"), (d.get("roofline") or {}).get("avg_phase2_ms"), "rest", (d.get("phase_split_ms") or {}).get("exchange_and_gaps"), "check", (d.get("check") or {}).get("ok"), "build", d.get("graph_build_s"), d.get("plan_build_s"))
PY
