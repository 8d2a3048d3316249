#!/usr/bin/env bash
# GPU box, round 3 session L: Louvain with hub hash table (parity, RMAT-22 / RMAT-26 timing, kernel stats)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "louvain" 2>&1 | tail -4
for hub in hash sort; do
  CUGRAPH_AMD_LOUVAIN_HUB=$hub timeout 600 python bench_louvain.py --scale 22 --cpu-scale 18 --out "$O/r3l_louvain_s22_$hub.json" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hub=$hub s22', d['value'], d['seconds_all'], 'frac', d['roofline']['frac'], 'check', d['check']['ok'])"
done
for hub in hash sort; do
  CUGRAPH_AMD_LOUVAIN_HUB=$hub CUGRAPH_AMD_LOUVAIN_TRACE=1 timeout 600 python bench_louvain.py --scale 26 --cpu-scale 0 --repeats 3 --out "$O/r3l_louvain_s26_$hub.json" 2>&1 | grep -E "\[louvain\] [0-9]|\"value\"" | tail -5 | cut -c1-160
done
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/prof_r3l_louvain"; timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_r3l_louvain" -o run -- python "$R/bench_louvain.py" --scale 22 --cpu-scale 0 --repeats 2 > "$O/r3l_louvain_prof.log" 2>&1
python "$R/tools/rocpd_summary.py" "$O/prof_r3l_louvain" > "$O/r3l_louvain_s22_rocprofv3_summary.txt" 2>&1; find "$O/prof_r3l_louvain" -name "*.db" -delete
head -22 "$O/r3l_louvain_s22_rocprofv3_summary.txt" | cut -c1-150
