#!/usr/bin/env bash
# GPU box, round 5 session r: the multi-GPU PageRank's exchange in two chunks -- bit identity first, then everything PageRank, then what it costs
# with the ranks on ONE GPU (plumbing: there is no wire to hide here)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_mg_capi.py -m gpu -q -x -k "two_chunk" 2>&1 | tail -40 | tee "$O/r5r_new.log"
timeout 900 python -m pytest tests/test_mg_capi.py tests/test_mg.py tests/test_reference_c_tests.py tests/test_gpu_parity.py tests/test_c_conformance.py -m gpu -q -k "(pagerank or conformance) and not two_chunk" 2>&1 | tail -12 | tee "$O/r5r_pagerank.log"
one() { timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) "$@"; }
for ov in 0 30; do
  CUGRAPH_AMD_MG_OVERLAP=$ov CUGRAPH_AMD_MG_TEST_SINGLE_GPU=1 timeout 600 python bench.py --gpus 2 --scale 24 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null > "$O/r5r_ipc2_s24_ov$ov.json"
  CUGRAPH_AMD_MG_OVERLAP=$ov CUGRAPH_AMD_MG_PUSH_SELF=1 one bench.py --gpus 2 --scale 26 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null > "$O/r5r_ipc1self_s26_ov$ov.json"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5r_ipc*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
        print(f.split("/")[-1], d["ms_per_step"], "p1", r.get("avg_phase1_ms"), "p2", r.get("avg_phase2_ms"), "check", (d.get("check") or {}).get("ok"))
    except Exception as e:
        print(f, "ERR", e)
PY
