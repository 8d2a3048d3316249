#!/usr/bin/env bash
# GPU box: traversal bench lines (BASELINE.json config 3) + rocprofv3 kernel summary
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
TAG=${TAG:-r2}
timeout 600 python bench_traversal.py --scale 24 --roots 64 --weights int --predecessors --out "$O/${TAG}_traversal_s24_int.json" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python bench_traversal.py --scale 24 --roots 64 --weights unit --no-cpu-baseline --out "$O/${TAG}_traversal_s24_unit.json" 2>&1 | grep -v amdgpu.ids | tail -1
timeout 600 python bench_traversal.py --scale 24 --roots 64 --symmetric --no-sssp --no-cpu-baseline --out "$O/${TAG}_traversal_s24_sym.json" 2>&1 | grep -v amdgpu.ids | tail -1
if [ "${PROF:-1}" = "1" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof_${TAG}_trav" -- python "$R/bench_traversal.py" --scale 24 --roots 8 --weights int --no-cpu-baseline > "$O/prof_${TAG}_trav.log" 2>&1
  cd "$R"; python tools/rocpd_summary.py "$O/prof_${TAG}_trav" > "$O/${TAG}_traversal_s24_rocprofv3_summary.txt" 2>&1 || true
  head -30 "$O/${TAG}_traversal_s24_rocprofv3_summary.txt"
fi
