#!/usr/bin/env bash
# GPU box: library variants of gpurun_libs/ (LIBS="a b ...") compared on the driver's command line in FRESH processes, alternating (REPS rounds): a process
# draws its own placement, so a variant is judged by its mean over the rounds.  Output gpurun_out/${TAG}_ab_fresh.txt
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
cp cugraph_amd/lib/libcugraph_c.so /tmp/orig.so
out="$O/${TAG:-ab}_ab_fresh.txt"; : > "$out"
for rep in $(seq 1 ${REPS:-8}); do for lib in ${LIBS:-base}; do
  cp "gpurun_libs/$lib.so" cugraph_amd/lib/libcugraph_c.so
  timeout 200 python bench.py --gpus 1 --scale ${SCALE:-26} --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-check ${BENCH_EXTRA:-} 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('rep $rep lib $lib ms_per_step', d['ms_per_step'], 'frac', r['frac'], 'phase1', r.get('avg_phase1_ms'), 'phase2', r.get('avg_phase2_ms'))" | tee -a "$out"
done; done
cp /tmp/orig.so cugraph_amd/lib/libcugraph_c.so
python - "$out" <<'PY' | tee -a "$out"
import sys,re,collections
a=collections.OrderedDict()
for l in open(sys.argv[1]):
    m=re.search(r"lib (\S+) ms_per_step ([\d.]+) frac [\d.]+ phase1 ([\d.]+) phase2 ([\d.]+)",l)
    if m: a.setdefault(m.group(1),[]).append(tuple(float(m.group(i)) for i in (2,3,4)))
for k,v in a.items():
    n=len(v); ms=[x[0] for x in v]
    print(f"{k}: n {n} mean {sum(ms)/n:.4f} min {min(ms):.4f} max {max(ms):.4f}  phase1 {sum(x[1] for x in v)/n:.4f} phase2 {sum(x[2] for x in v)/n:.4f}")
PY
