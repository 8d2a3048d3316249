#!/usr/bin/env bash
# round 5, session i: Louvain with the chunk kernel's per-level preparation (parity: oracle tests + fixtures; timing at RMAT-22 / 26)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "louvain" --durations=3 2>&1 | tail -8 | tee "$O/r5i_tests.log"
for i in 1 2; do timeout 300 python bench_louvain.py --scale 22 --cpu-scale 0 --repeats 3 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("louvain s22", d["value"], d["unit"], "frac", d["roofline"]["frac"], "check", (d.get("check") or {}).get("ok"))' | tee -a "$O/r5i_louvain.txt"; done
timeout 600 python bench_louvain.py --scale 26 --cpu-scale 0 --repeats 2 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("louvain s26", d["value"], d["unit"], "frac", d["roofline"]["frac"], "check", (d.get("check") or {}).get("ok"))' | tee -a "$O/r5i_louvain.txt"
