#!/usr/bin/env bash
# GPU box: the session the committed round-5 numbers come from (outputs gpurun_out/r5z_*; copy to profiles/).  Most important first.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${TAG:-r5z}
# the plain-C conformance binary as the first GPU processes of the lease (the one-off memory fault of round 4: DESIGN.md section 7)
timeout 400 python tools/conformance_loop.py 50 2>&1 | tail -3 | tee "$O/${TAG}_conformance_loop.log"
timeout 1500 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -20 | tee "$O/${TAG}_pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$O/${TAG}_smoke.log"
# counter traffic first: the bench lines below quote it (same session, same source hash)
timeout 1500 python tools/traffic_collect.py > "$O/${TAG}_traffic.log" 2>&1; tail -7 "$O/${TAG}_traffic.log" | cut -c1-300
cp "$O/traffic_latest.json" profiles/traffic_latest.json
timeout 900 python bench.py 2>"$O/${TAG}_bench.err" > "$O/${TAG}_bench_s26.json"; echo "bench rc=$?"; cut -c1-200 "$O/${TAG}_bench_s26.json"
for sc in 22 24; do timeout 300 python bench.py --scale $sc --no-extras --cpu-scale 20 2>/dev/null > "$O/${TAG}_bench_s$sc.json"; cut -c1-120 "$O/${TAG}_bench_s$sc.json"; done
( cd /tmp && export TMPDIR=/tmp
rm -rf "$O/prof_$TAG"; timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_$TAG/pr" -o run -- python "$R/bench.py" --steps 20 --warmup 1 --no-check --no-cpu-baseline --no-extras > "$O/prof_$TAG.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_$TAG/trav" -o run -- python "$R/bench_traversal.py" --scale 24 --weights int --roots 8 --no-cpu-baseline --no-check >> "$O/prof_$TAG.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_$TAG/louv" -o run -- python "$R/bench_louvain.py" --scale 22 --cpu-scale 0 --repeats 3 >> "$O/prof_$TAG.log" 2>&1
python "$R/tools/rocpd_summary.py" "$O/prof_$TAG/pr" > "$O/${TAG}_s26_rocprofv3_summary.txt" 2>&1
python "$R/tools/rocpd_summary.py" --overlap "k_tiled_phase1" "k_tiled_phase2" "$O/prof_$TAG/pr" 2>&1 | head -12 >> "$O/${TAG}_s26_rocprofv3_summary.txt"
python "$R/tools/rocpd_summary.py" "$O/prof_$TAG/trav" > "$O/${TAG}_traversal_s24_rocprofv3_summary.txt" 2>&1
python "$R/tools/rocpd_summary.py" "$O/prof_$TAG/louv" > "$O/${TAG}_louvain_s22_rocprofv3_summary.txt" 2>&1
find "$O/prof_$TAG" -name "*.db" -delete
head -8 "$O/${TAG}_s26_rocprofv3_summary.txt" | cut -c1-150 )
timeout 600 python bench_traversal.py --scale 24 --weights int --out "$O/${TAG}_traversal_s24_int.json" > /dev/null 2>&1
timeout 600 python bench_traversal.py --scale 24 --weights unit --out "$O/${TAG}_traversal_s24_unit.json" > /dev/null 2>&1
timeout 600 python bench_traversal.py --scale 24 --symmetric --no-sssp --out "$O/${TAG}_traversal_s24_sym.json" > /dev/null 2>&1
timeout 600 python bench_louvain.py --scale 22 --out "$O/${TAG}_louvain_s22.json" > /dev/null 2>&1
timeout 900 python bench_louvain.py --scale 26 --cpu-scale 0 --repeats 2 --out "$O/${TAG}_louvain_s26.json" > /dev/null 2>&1
# multi-GPU entry points with the ranks on this one GPU: one rank through the partitioned paths (overhead against the lines above), plumbing at 2 / 8
one() { timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) "$@"; }
one bench.py --gpus 2 --scale 24 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null > "$O/${TAG}_ipc1_s24.json"
one bench.py --gpus 2 --scale 26 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null > "$O/${TAG}_ipc1_s26.json"
for w in 2 8; do CUGRAPH_AMD_MG_TEST_SINGLE_GPU=1 timeout 600 python bench.py --gpus $w --scale 24 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null > "$O/${TAG}_ipc${w}_s24.json"; done
timeout 600 python bench_traversal.py --partitioned --transport ipc --scale 24 --weights int --roots 16 2>/dev/null | grep "^{" > "$O/${TAG}_part_ipc1_s24.json"
# the x exchange in two chunks (opt-in, CUGRAPH_AMD_MG_OVERLAP): one rank pushing to ITSELF at RMAT-26, one chunk against two; then a kernel trace of
# the two-chunk run (one process, no launcher) with the push kernels set against phase 2
for ov in 0 30; do CUGRAPH_AMD_MG_OVERLAP=$ov CUGRAPH_AMD_MG_PUSH_SELF=1 one bench.py --gpus 2 --scale 26 --steps 20 --warmup 3 --no-cpu-baseline --no-check 2>/dev/null > "$O/${TAG}_ipc1self_s26_ov$ov.json"; done
( cd /tmp && export TMPDIR=/tmp
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29917 CUGRAPH_AMD_MG_OVERLAP=30 CUGRAPH_AMD_MG_PUSH_SELF=1 timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_$TAG/mgovl" -o run -- \
  python "$R/bench.py" --gpus 2 --scale 24 --steps 10 --warmup 2 --no-cpu-baseline --no-check > "$O/prof_${TAG}_mgovl.log" 2>&1
python "$R/tools/rocpd_summary.py" "$O/prof_$TAG/mgovl" 2>&1 | head -14 > "$O/${TAG}_mg_two_chunk_rocprofv3_summary.txt"
python "$R/tools/rocpd_summary.py" --overlap "k_mgc_push" "k_tiled_phase2" "$O/prof_$TAG/mgovl" 2>&1 | head -14 >> "$O/${TAG}_mg_two_chunk_rocprofv3_summary.txt"
find "$O/prof_$TAG" -name "*.db" -delete
tail -6 "$O/${TAG}_mg_two_chunk_rocprofv3_summary.txt" | cut -c1-160 )
CUGRAPH_AMD_MG_TEST_SINGLE_GPU=1 timeout 900 python bench_louvain.py --gpus 2 --scale 22 --repeats 2 --out "$O/${TAG}_louvain_s22_ranks2.json" > /dev/null 2>&1
timeout 900 python bench_traversal.py --scale 26 --symmetric --roots 16 --no-sssp --no-cpu-baseline --out "$O/${TAG}_traversal_s26_sym.json" > /dev/null 2>&1
python - <<'PY'
import json,glob,os
tag=os.environ.get("TAG","r5z")
for f in sorted(glob.glob(f"gpurun_out/{tag}_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    r=d.get("roofline") or {}
    print(f.split("/")[-1], d.get("value"), d.get("ms_per_step"), "frac", r.get("frac"), "traffic", r.get("traffic"), "check", (d.get("check") or {}).get("ok"),
          {k: (v.get("value"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("traffic"), (v.get("check") or {}).get("ok")) for k, v in (d.get("extra") or {}).items() if isinstance(v, dict)},
          {k: (d[k].get("mean_ms"), (d[k].get("roofline") or {}).get("frac"), (d[k].get("roofline") or {}).get("traffic"), (d[k].get("check") or {}).get("ok")) for k in ("bfs","sssp") if k in d})
PY
