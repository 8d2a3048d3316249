#!/usr/bin/env bash
# GPU box, round 5 session s: the two-chunk exchange with smoothed tails -- bit identity at 1 .. 4 ranks, one rank pushing to itself at RMAT-26 both ways
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_mg_capi.py -m gpu -q -k "two_chunk" 2>&1 | tail -30 | tee "$O/r5s_new.log"
one() { timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) "$@"; }
for ov in 0 30 30p; do
  if [ $ov = 30p ]; then export CUGRAPH_AMD_MG_OVERLAP_PLAIN_TAIL=1; v=30; else unset CUGRAPH_AMD_MG_OVERLAP_PLAIN_TAIL; v=$ov; fi
  CUGRAPH_AMD_MG_OVERLAP=$v CUGRAPH_AMD_MG_PUSH_SELF=1 one bench.py --gpus 2 --scale 26 --steps 20 --warmup 3 --no-cpu-baseline --no-check 2>/dev/null > "$O/r5s_ipc1self_s26_ov$ov.json"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5s_ipc*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
        print(f.split("/")[-1], d["ms_per_step"], "p1", r.get("avg_phase1_ms"), "p2", r.get("avg_phase2_ms"), "check", (d.get("check") or {}).get("ok"))
    except Exception as e:
        print(f, "ERR", e)
PY
