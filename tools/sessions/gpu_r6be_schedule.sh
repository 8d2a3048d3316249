R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; export HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r6be_p1_schedule_s26.txt; : > $out
for rep in 1 2 3 4 5; do for cfg in "default" "CUGRAPH_AMD_TP_STATIC_FRAC=0.3" "CUGRAPH_AMD_TP_STATIC_FRAC=0.5" "CUGRAPH_AMD_TP_CHUNK=64" "CUGRAPH_AMD_TP_CHUNK=16"; do
  if [ "$cfg" = default ]; then e=""; else e="$cfg"; fi
  env $e timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-check 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('rep $rep cfg $cfg ms_per_step', d['ms_per_step'], 'phase1', r.get('avg_phase1_ms'), 'phase2', r.get('avg_phase2_ms'))" | tee -a $out
done; done
python - $out <<'PY' | tee -a $out
import sys,re,collections
a=collections.OrderedDict()
for l in open(sys.argv[1]):
    m=re.search(r"cfg (\S+) ms_per_step ([\d.]+) phase1 ([\d.]+)",l)
    if m: a.setdefault(m.group(1),[]).append((float(m.group(2)),float(m.group(3))))
for k,v in a.items(): print(f"{k}: n {len(v)} mean {sum(x[0] for x in v)/len(v):.4f} phase1 {sum(x[1] for x in v)/len(v):.4f}")
PY
