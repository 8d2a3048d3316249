#!/usr/bin/env bash
# GPU box, round 2 experiment 1: chunk-length sweep for phase 1 + the one-rank multi-GPU engine on the same graph
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
L="$O/exp1.log"; : > "$L"
fmt='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print("ms/step", d["ms_per_step"], "p1", r.get("avg_phase1_ms"), "p2", r.get("avg_phase2_ms"), "frac", r["frac"], "build", d.get("graph_build_s"), d.get("plan_build_s"), "check", d.get("check"))
    elif l.startswith("[tiled"): print(l.strip())'
run() { echo "== $*" >> "$L"; env "$@" CUGRAPH_AMD_TILED_DEBUG=1 timeout 300 python bench.py --scale ${SCALE:-26} --steps 20 --warmup 5 --no-cpu-baseline $EXTRA 2>&1 | grep -v amdgpu.ids | python -c "$fmt" >> "$L" 2>&1; }
EXTRA="" run A=0
EXTRA="--no-check" run CUGRAPH_AMD_TP_CHUNK_BIG=32
EXTRA="--no-check" run CUGRAPH_AMD_TP_CHUNK_BIG=64
EXTRA="--no-check" run CUGRAPH_AMD_TP_CHUNK_BIG=128 CUGRAPH_AMD_TP_CHUNK_BIG_FRAC=0.7
EXTRA="--no-check" run CUGRAPH_AMD_PAGERANK_DENSE_COLUMNS=1
echo "== mg one rank" >> "$L"
CUGRAPH_AMD_TILED_DEBUG=1 bash tools/gpu_mgdebug.sh >> "$L" 2>&1
cat "$L"
