#!/usr/bin/env bash
# GPU box: after a change to the PageRank kernels only -- its parity tests, its counter traffic (the other workloads' entries of
# profiles/traffic_latest.json keep their hashes), the bench line and the kernel summary again (outputs gpurun_out/<TAG>_*)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${TAG:-r5zb}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_mg_capi.py tests/test_mg.py tests/test_c_conformance.py tests/test_reference_c_tests.py -m gpu -q -k "pagerank or conformance or reference" 2>&1 | tail -4 | tee "$O/${TAG}_pytest_pagerank.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee "$O/${TAG}_smoke.log"
# re-measure the PageRank entry and merge it into the committed counter file (the traversal / Louvain entries are untouched: their sources did not change)
cp profiles/traffic_latest.json "$O/traffic_before.json"
timeout 600 python tools/traffic_collect.py pagerank_s26 > "$O/${TAG}_traffic.log" 2>&1; tail -3 "$O/${TAG}_traffic.log" | cut -c1-300
python - <<'PY'
import json
old = json.load(open("gpurun_out/traffic_before.json")); new = json.load(open("gpurun_out/traffic_latest.json"))
old["entries"]["pagerank_s26"] = new["entries"]["pagerank_s26"]
old["source_hash"] = new["source_hash"]
old["group_hashes"]["pagerank"] = new["group_hashes"]["pagerank"]
for g in ("traversal", "louvain"):
    assert old["group_hashes"][g] == new["group_hashes"][g], (g, "its sources changed too: run tools/gpu_final5.sh")
json.dump(old, open("gpurun_out/traffic_latest.json", "w"), indent=1); open("gpurun_out/traffic_latest.json", "a").write("\n")
json.dump(old, open("profiles/traffic_latest.json", "w"), indent=1); open("profiles/traffic_latest.json", "a").write("\n")
print("pagerank_s26", old["entries"]["pagerank_s26"].get("hbm_bytes"))
PY
timeout 900 python bench.py 2>"$O/${TAG}_bench.err" > "$O/${TAG}_bench_s26.json"; echo "bench rc=$?"; cut -c1-260 "$O/${TAG}_bench_s26.json"
for sc in 22 24; do timeout 300 python bench.py --scale $sc --no-extras --cpu-scale 20 2>/dev/null > "$O/${TAG}_bench_s$sc.json"; cut -c1-140 "$O/${TAG}_bench_s$sc.json"; done
( cd /tmp && export TMPDIR=/tmp
rm -rf "$O/prof_$TAG"; timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_$TAG/pr" -o run -- python "$R/bench.py" --steps 20 --warmup 1 --no-check --no-cpu-baseline --no-extras > "$O/prof_$TAG.log" 2>&1
python "$R/tools/rocpd_summary.py" "$O/prof_$TAG/pr" > "$O/${TAG}_s26_rocprofv3_summary.txt" 2>&1
python "$R/tools/rocpd_summary.py" --overlap "k_tiled_phase1" "k_tiled_phase2" "$O/prof_$TAG/pr" 2>&1 | head -12 >> "$O/${TAG}_s26_rocprofv3_summary.txt"
find "$O/prof_$TAG" -name "*.db" -delete
grep "k_tiled_phase" "$O/${TAG}_s26_rocprofv3_summary.txt" | cut -c1-150 )
