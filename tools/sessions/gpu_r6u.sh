#!/usr/bin/env bash
# GPU box: is the gain of the placement trials the placement, or the longer GPU activity before the timed region (clocks)?  Fresh processes, alternating:
#   A  no trials, --warmup 5 (the driver's line)     B  no trials, --warmup 100     C  8 trials, --warmup 5     D  8 trials, --warmup 100
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > "$O/r6u_warmup_or_placement.txt"
for rep in 1 2 3 4 5; do for cfg in "A 1 5" "B 1 100" "C 8 5" "D 8 100"; do set -- $cfg
  CUGRAPH_AMD_PR_PLACEMENT_TRIALS=$2 timeout 300 python bench.py --gpus 1 --steps 20 --warmup $3 --no-extras --no-cpu-baseline --no-check 2>/dev/null | grep ms_per_step \
   | sed -E "s/.*\"ms_per_step\": ([0-9.]+).*avg_phase1_ms\": ([0-9.]+), \"avg_phase2_ms\": ([0-9.]+).*/rep $rep cfg $1 (trials $2 warmup $3): ms_per_step \1 phase1 \2 phase2 \3/" | cut -c1-160 >> "$O/r6u_warmup_or_placement.txt"
done; done
python - <<'PY'
import re,collections
d=collections.defaultdict(list)
for l in open("gpurun_out/r6u_warmup_or_placement.txt"):
    m=re.match(r"rep \d+ cfg (\S+) \((.*?)\): ms_per_step (\S+)",l)
    if m: d[m.group(1)+" "+m.group(2)].append(float(m.group(3)))
for k,v in sorted(d.items()): print(k,"mean %.4f min %.4f max %.4f |"%(sum(v)/len(v),min(v),max(v)), " ".join("%.4f"%x for x in v))
PY
