#!/usr/bin/env bash
# round 4, session p: full GPU suite with durations (the RMAT-26 Louvain fixture is new), window memory kinds for the PageRank gather
# window (cached / fine-grained / uncached) at one and two ranks, bench --gpus 2 with the bring-up self-check, conformance reruns
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 2>&1 | tail -45 | tee "$O/r4p_suite.log"
one() { timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) bench.py --gpus 2 "$@"; }
for kind in cached finegrained uncached; do
  CUGRAPH_AMD_COMM_WINDOWS=$kind one --scale 24 --steps 20 --warmup 3 --no-cpu-baseline 2>"$O/r4p_ipc1_s24_$kind.err" > "$O/r4p_ipc1_s24_$kind.json"; echo "one rank $kind rc=$?"
  CUGRAPH_AMD_COMM_WINDOWS=$kind CUGRAPH_AMD_MG_TEST_SINGLE_GPU=1 timeout 600 python bench.py --gpus 2 --scale 24 --steps 10 --warmup 2 --no-cpu-baseline 2>"$O/r4p_ipc2_s24_$kind.err" > "$O/r4p_ipc2_s24_$kind.json"; echo "two ranks $kind rc=$?"; tail -2 "$O/r4p_ipc2_s24_$kind.err" | cut -c1-300
done
for i in 1 2 3; do timeout 300 python -m pytest tests/test_c_conformance.py -m gpu -x -q 2>&1 | tail -2; done | tee "$O/r4p_conformance.log"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4p_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], "ms/step", d.get("ms_per_step"), "p1/p2", (d.get("phase_split_ms") or {}).get("phase1"), (d.get("phase_split_ms") or {}).get("phase2"), "rest", (d.get("phase_split_ms") or {}).get("exchange_and_gaps"), "check", (d.get("check") or {}).get("ok"), "link", (d.get("config") or {}).get("link_selftest"), (d.get("config") or {}).get("transport_note"))
PY
