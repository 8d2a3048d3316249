#!/usr/bin/env bash
# GPU box: the session the committed round-6 numbers come from (outputs gpurun_out/r6z_*; copy to profiles/).  Most important first.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${TAG:-r6z}
timeout 400 python tools/conformance_loop.py 30 2>&1 | tail -3 | tee "$O/${TAG}_conformance_loop.log"
timeout 1800 python -m pytest tests -m gpu -q --durations=10 2>&1 | tail -18 | tee "$O/${TAG}_pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$O/${TAG}_smoke.log"
# counter traffic first: the bench lines below quote it (same session, same source hash)
timeout 1500 python tools/traffic_collect.py > "$O/${TAG}_traffic.log" 2>&1; tail -7 "$O/${TAG}_traffic.log" | cut -c1-300
cp "$O/traffic_latest.json" profiles/traffic_latest.json
timeout 900 python bench.py 2>"$O/${TAG}_bench.err" > "$O/${TAG}_bench_s26.json"; echo "bench rc=$?"; cut -c1-200 "$O/${TAG}_bench_s26.json"
# the driver's own command line, three fresh processes (the spread between processes: DESIGN.md section 3.1, round 6)
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-check 2>/dev/null > "$O/${TAG}_bench_s26_rep$i.json"; cut -c1-130 "$O/${TAG}_bench_s26_rep$i.json"; done
# the same without the plan's placement tuning (what rounds 1-5 and the first r6z session measured)
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --placements 1 --no-extras --no-cpu-baseline --no-check 2>/dev/null > "$O/${TAG}_bench_s26_untuned_rep$i.json"; cut -c1-130 "$O/${TAG}_bench_s26_untuned_rep$i.json"; done
for sc in 22 23 24 25; do timeout 300 python bench.py --scale $sc --no-extras --cpu-scale 20 2>/dev/null > "$O/${TAG}_bench_s$sc.json"; cut -c1-120 "$O/${TAG}_bench_s$sc.json"; done
( cd /tmp && export TMPDIR=/tmp
rm -rf "$O/prof_$TAG"; timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_$TAG/pr" -o run -- python "$R/bench.py" --steps 20 --warmup 1 --placements 1 --no-check --no-cpu-baseline --no-extras > "$O/prof_$TAG.log" 2>&1  # (untuned: every launch of the trace is a launch of the ONE placement the line's averages are about)
timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_$TAG/trav" -o run -- python "$R/bench_traversal.py" --scale 24 --weights int --roots 8 --no-cpu-baseline --no-check >> "$O/prof_$TAG.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_$TAG/louv" -o run -- python "$R/bench_louvain.py" --scale 22 --cpu-scale 0 --repeats 3 >> "$O/prof_$TAG.log" 2>&1
python "$R/tools/rocpd_summary.py" "$O/prof_$TAG/pr" > "$O/${TAG}_s26_rocprofv3_summary.txt" 2>&1
python "$R/tools/rocpd_summary.py" "$O/prof_$TAG/trav" > "$O/${TAG}_traversal_s24_rocprofv3_summary.txt" 2>&1
python "$R/tools/rocpd_summary.py" "$O/prof_$TAG/louv" > "$O/${TAG}_louvain_s22_rocprofv3_summary.txt" 2>&1
find "$O/prof_$TAG" -name "*.db" -delete
head -8 "$O/${TAG}_s26_rocprofv3_summary.txt" | cut -c1-150 )
TAG=$TAG bash tools/gpu_prof_tuned.sh  # the driver's own line (tuned plan) under rocprofv3: the trace's last 20 launches per kernel = the timed region
timeout 600 python bench_traversal.py --scale 24 --weights int --out "$O/${TAG}_traversal_s24_int.json" > /dev/null 2>&1
timeout 600 python bench_traversal.py --scale 24 --weights unit --out "$O/${TAG}_traversal_s24_unit.json" > /dev/null 2>&1
CUGRAPH_AMD_SSSP_FILTER=0 timeout 600 python bench_traversal.py --scale 24 --weights int --roots 16 --no-cpu-baseline --out "$O/${TAG}_traversal_s24_int_nofilter.json" > /dev/null 2>&1
timeout 600 python bench_traversal.py --scale 24 --symmetric --no-sssp --out "$O/${TAG}_traversal_s24_sym.json" > /dev/null 2>&1
timeout 600 python bench_louvain.py --scale 22 --out "$O/${TAG}_louvain_s22.json" > /dev/null 2>&1
timeout 900 python bench_louvain.py --scale 26 --cpu-scale 0 --repeats 2 --out "$O/${TAG}_louvain_s26.json" > /dev/null 2>&1
# multi-GPU entry points with the ranks on this one GPU: one rank through the partitioned paths (overhead against the lines above), plumbing at 2 / 8, both layouts
one() { timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) "$@"; }
one bench.py --gpus 2 --scale 24 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null > "$O/${TAG}_ipc1_s24.json"
one bench.py --gpus 2 --scale 26 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null > "$O/${TAG}_ipc1_s26.json"
one bench.py --gpus 2 --scale 26 --steps 20 --warmup 3 --no-cpu-baseline --layout 2d 2>/dev/null > "$O/${TAG}_ipc1_s26_2d.json"
for lay in 1d 2d; do for w in 2 8; do CUGRAPH_AMD_MG_TEST_SINGLE_GPU=1 timeout 600 python bench.py --gpus $w --scale 24 --steps 10 --warmup 2 --no-cpu-baseline --layout $lay 2>/dev/null > "$O/${TAG}_ipc${w}_s24_$lay.json"; done; done
timeout 600 python bench_traversal.py --partitioned --transport ipc --scale 24 --weights int --roots 16 2>/dev/null | grep "^{" > "$O/${TAG}_part_ipc1_s24.json"
CUGRAPH_AMD_MG_TEST_SINGLE_GPU=1 timeout 900 python bench_louvain.py --gpus 2 --scale 22 --repeats 2 --out "$O/${TAG}_louvain_s22_ranks2.json" > /dev/null 2>&1
timeout 900 python bench_traversal.py --scale 26 --symmetric --roots 16 --no-sssp --no-cpu-baseline --out "$O/${TAG}_traversal_s26_sym.json" > /dev/null 2>&1
python - <<'PY'
import json,glob,os
tag=os.environ.get("TAG","r6z")
for f in sorted(glob.glob(f"gpurun_out/{tag}_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    r=d.get("roofline") or {}
    print(f.split("/")[-1], d.get("value"), d.get("ms_per_step"), "frac", r.get("frac"), "traffic", r.get("traffic"), "check", (d.get("check") or {}).get("ok"),
          {k: (v.get("value"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("traffic"), (v.get("check") or {}).get("ok")) for k, v in (d.get("extra") or {}).items() if isinstance(v, dict)},
          {k: (d[k].get("mean_ms"), (d[k].get("roofline") or {}).get("frac"), (d[k].get("roofline") or {}).get("traffic"), (d[k].get("check") or {}).get("ok")) for k in ("bfs","sssp") if k in d})
PY
