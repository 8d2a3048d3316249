#!/usr/bin/env bash
# GPU box: the session the committed round-3 numbers come from (outputs gpurun_out/r3z_*; copy to profiles/)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
TAG=${TAG:-r3z}
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee "$O/${TAG}_pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$O/${TAG}_smoke.log"
# counter traffic first: the bench lines below quote it (same session, same source hash)
timeout 1500 python tools/traffic_collect.py > "$O/${TAG}_traffic.log" 2>&1; tail -7 "$O/${TAG}_traffic.log" | cut -c1-300
cp "$O/traffic_latest.json" profiles/traffic_latest.json
timeout 900 python bench.py 2>"$O/${TAG}_bench.err" > "$O/${TAG}_bench_s26.json"; echo "bench rc=$?"; cut -c1-200 "$O/${TAG}_bench_s26.json"
for sc in 22 24; do timeout 300 python bench.py --scale $sc --no-extras --cpu-scale 20 2>/dev/null > "$O/${TAG}_bench_s$sc.json"; cut -c1-120 "$O/${TAG}_bench_s$sc.json"; done
timeout 600 python bench_traversal.py --scale 24 --weights int --out "$O/${TAG}_traversal_s24_int.json" > /dev/null 2>&1
timeout 600 python bench_traversal.py --scale 24 --weights unit --out "$O/${TAG}_traversal_s24_unit.json" > /dev/null 2>&1
timeout 600 python bench_traversal.py --scale 24 --symmetric --no-sssp --out "$O/${TAG}_traversal_s24_sym.json" > /dev/null 2>&1
timeout 900 python bench_traversal.py --scale 26 --symmetric --roots 16 --no-sssp --no-cpu-baseline --out "$O/${TAG}_traversal_s26_sym.json" > /dev/null 2>&1
timeout 600 python bench_louvain.py --scale 22 --out "$O/${TAG}_louvain_s22.json" > /dev/null 2>&1
timeout 900 python bench_louvain.py --scale 26 --cpu-scale 0 --repeats 2 --out "$O/${TAG}_louvain_s26.json" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/prof_$TAG"; timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_$TAG/pr" -o run -- python "$R/bench.py" --steps 20 --warmup 1 --no-check --no-cpu-baseline --no-extras > "$O/prof_$TAG.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_$TAG/trav" -o run -- python "$R/bench_traversal.py" --scale 24 --weights int --roots 8 --no-cpu-baseline --no-check >> "$O/prof_$TAG.log" 2>&1
python "$R/tools/rocpd_summary.py" "$O/prof_$TAG/pr" > "$O/${TAG}_s26_rocprofv3_summary.txt" 2>&1
python "$R/tools/rocpd_summary.py" "$O/prof_$TAG/trav" > "$O/${TAG}_traversal_s24_rocprofv3_summary.txt" 2>&1
find "$O/prof_$TAG" -name "*.db" -delete
head -8 "$O/${TAG}_s26_rocprofv3_summary.txt" | cut -c1-150
cd "$R"
# the N > 1 launch line with a ONE-rank RCCL process group (device tensors, library primitives, nccl collectives), both layouts
for lay in 1d 2d; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --layout $lay --scale 24 --steps 10 --warmup 2 --cpu-scale 18 2>"$O/${TAG}_nccl1_$lay.err" > "$O/${TAG}_nccl1_$lay.json"; echo "nccl world-1 layout $lay rc=$?"; cut -c1-260 "$O/${TAG}_nccl1_$lay.json"; tail -2 "$O/${TAG}_nccl1_$lay.err" | cut -c1-200
done
# the partitioned traversals with one rank over RCCL (BFS direction-optimising; SSSP)
timeout 600 python bench_traversal.py --partitioned --scale 24 --weights int --roots 16 2>/dev/null | grep "^{" > "$O/${TAG}_partitioned_s24.json"; cut -c1-400 "$O/${TAG}_partitioned_s24.json"
# the N > 1 line at the full size with one rank (plan construction through the library primitives at 2^30 edges)
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 2 --layout 1d --scale 26 --steps 10 --warmup 2 --cpu-scale 18 2>"$O/${TAG}_nccl1_1d_s26.err" > "$O/${TAG}_nccl1_1d_s26.json"; echo "nccl world-1 scale 26 rc=$?"; cut -c1-200 "$O/${TAG}_nccl1_1d_s26.json"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3z_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    r=d.get("roofline") or {}
    print(f.split("/")[-1], d.get("value"), d.get("ms_per_step"), "frac", r.get("frac"), "traffic", r.get("traffic"), "check", (d.get("check") or {}).get("ok"),
          {k: (v.get("value"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("traffic"), (v.get("check") or {}).get("ok")) for k, v in (d.get("extra") or {}).items() if isinstance(v, dict)},
          {k: (d[k].get("mean_ms"), d[k]["roofline"]["frac"], d[k]["roofline"].get("traffic"), (d[k].get("check") or {}).get("ok")) for k in ("bfs","sssp") if k in d})
PY
