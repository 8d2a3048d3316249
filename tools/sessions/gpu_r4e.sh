#!/usr/bin/env bash
# round 4, session e: bench.py --gpus N on the library's communicator: one rank at RMAT-24 / 26 (the partitioned path against the
# single-GPU entry point on the same box), 2 and 4 ranks sharing the GPU at RMAT-22 (plumbing + check)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
one() { timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --gpus 2 "$@"; }
for sc in 24 26; do
  one --scale $sc --steps 20 --warmup 3 --no-cpu-baseline 2>"$O/r4e_ipc1_s$sc.err" > "$O/r4e_ipc1_s$sc.json"; echo "ipc one rank s$sc rc=$?"; cut -c1-300 "$O/r4e_ipc1_s$sc.json"; tail -2 "$O/r4e_ipc1_s$sc.err" | cut -c1-300
  timeout 600 python bench.py --scale $sc --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-check 2>/dev/null > "$O/r4e_sg_s$sc.json"; cut -c1-200 "$O/r4e_sg_s$sc.json"
done
CUGRAPH_AMD_PAGERANK_ALL_ROWS=1 one --scale 26 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null > "$O/r4e_ipc1_s26_allrows.json"; cut -c1-200 "$O/r4e_ipc1_s26_allrows.json"
for w in 2 4; do
  CUGRAPH_AMD_MG_TEST_SINGLE_GPU=1 timeout 600 python bench.py --gpus $w --scale 22 --steps 10 --warmup 2 --cpu-scale 18 2>"$O/r4e_ipc${w}_s22.err" > "$O/r4e_ipc${w}_s22.json"; echo "ipc $w ranks on one GPU rc=$?"; cut -c1-400 "$O/r4e_ipc${w}_s22.json"; tail -2 "$O/r4e_ipc${w}_s22.err" | cut -c1-300
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4e_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], "ms/step", d.get("ms_per_step"), "frac", (d.get("roofline") or {}).get("frac"), "p1/p2", (d.get("roofline") or {}).get("avg_phase1_ms"), (d.get("roofline") or {}).get("avg_phase2_ms"), "split", d.get("phase_split_ms"), "check", (d.get("check") or {}).get("ok"), "build", d.get("graph_build_s"), d.get("plan_build_s"))
PY
