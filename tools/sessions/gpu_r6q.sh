#!/usr/bin/env bash
# GPU box: phase-2 first-generation stagger (k_tiled_phase2, a.stagger) against off / other widths, alternated on one plan, several placements and scales.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
: > "$O/r6q_phase2_stagger.txt"
for sc in 26 26 26 25 25 24 24 23; do
  echo "== scale $sc (base = the library's default: 4 on grids >= 4096 workgroups, else 0)" >> "$O/r6q_phase2_stagger.txt"
  timeout 300 python tools/variant_ab.py $sc CUGRAPH_AMD_P2_STAGGER=0,2,4,6 3 2>&1 | grep -E "^rep|bit-identical" >> "$O/r6q_phase2_stagger.txt"
done
python - <<'PY'
import re,collections
d=collections.defaultdict(list); sc=None
for l in open("gpurun_out/r6q_phase2_stagger.txt"):
    if l.startswith("=="): sc=l.split()[2]; continue
    m=re.match(r"rep \d+ (\S+)\s+ms/iter (\S+) phase1 (\S+) phase2 (\S+)",l)
    if m: d[(sc,m.group(1))].append((float(m.group(2)),float(m.group(4))))
for k,v in sorted(d.items()):
    v.sort(); print(k, "median ms/iter %.4f  phase2 %.4f  n=%d"%(sorted(x[0] for x in v)[len(v)//2], sorted(x[1] for x in v)[len(v)//2], len(v)))
PY
