#!/usr/bin/env bash
# GPU box: placement trials of the PageRank plan (pagerank.hip: tune_placement) -- fresh processes of the driver's command line, alternating 1 (off) / 4 / 8 placements
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > "$O/r6t_placement_trials.txt"
for rep in 1 2 3 4 5 6; do for n in 1 4 8; do
  CUGRAPH_AMD_PR_PLACEMENT_TRIALS=$n CUGRAPH_AMD_PR_PLACEMENT_TRACE=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-check 2>&1 | grep -E "placement|ms_per_step" \
   | sed -E "s/.*\"ms_per_step\": ([0-9.]+).*avg_phase1_ms\": ([0-9.]+), \"avg_phase2_ms\": ([0-9.]+).*/rep $rep trials $n: bench ms_per_step \1 phase1 \2 phase2 \3/" | cut -c1-160 >> "$O/r6t_placement_trials.txt"
done; done
python - <<'PY'
import re,collections
d=collections.defaultdict(list)
for l in open("gpurun_out/r6t_placement_trials.txt"):
    m=re.match(r"rep \d+ trials (\d+): bench ms_per_step (\S+)",l)
    if m: d[int(m.group(1))].append(float(m.group(2)))
for k,v in sorted(d.items()): print("trials",k,"ms_per_step mean %.4f min %.4f max %.4f"%(sum(v)/len(v),min(v),max(v)), " ".join("%.4f"%x for x in v))
PY
