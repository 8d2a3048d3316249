#!/usr/bin/env bash
# GPU box: phase 2's first-generation stagger re-swept in fresh processes of the driver's line (after the priority change of phase 1)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; export HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r6bl_p2_stagger.txt; : > $out
for rep in 1 2 3 4 5 6; do for st in 6 0 3 10 16; do
  CUGRAPH_AMD_P2_STAGGER=$st timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('rep $rep stagger $st ms_per_step', d['ms_per_step'], 'phase1', r.get('avg_phase1_ms'), 'phase2', r.get('avg_phase2_ms'))" | tee -a $out
done; done
python - $out <<'PY' | tee -a $out
import sys,re,collections
a=collections.OrderedDict()
for l in open(sys.argv[1]):
    m=re.search(r"stagger (\d+) ms_per_step ([\d.]+) phase1 ([\d.]+) phase2 ([\d.]+)",l)
    if m: a.setdefault(m.group(1),[]).append((float(m.group(2)),float(m.group(4))))
for k,v in a.items(): print(f"stagger {k}: n {len(v)} mean {sum(x[0] for x in v)/len(v):.4f} phase2 {sum(x[1] for x in v)/len(v):.4f}")
PY
