#!/usr/bin/env bash
# round 6, session a: (1) the libraries of rounds 3, 4 and HEAD on ONE box, interleaved, each with its own bench.py (build/rel/<round> = git archive of the
# round's last commit, built in place) -- is the falling driver-clock headline a regression or the boxes?  (2) HEAD's phase-1 ablations (loads only, no stores)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
fmt='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); r=d["roofline"]; print("ms_per_step", d["ms_per_step"], "phase1", r.get("avg_phase1_ms"), "phase2", r.get("avg_phase2_ms"), "frac", r["frac"])'
: > "$O/r6a_release_ab.txt"
for rep in 1 2 3; do for t in r3 r4 head; do
  d="$R/build/rel/$t"; [ $t = head ] && d="$R"
  echo -n "rep $rep tree=$t " >> "$O/r6a_release_ab.txt"
  (cd "$d" && timeout 300 python bench.py --scale 26 --steps 20 --warmup 3 --no-cpu-baseline --no-check --no-extras 2>&1 | tee "$O/r6a_last_bench_$t.log" | tail -2 | python -c "$fmt") >> "$O/r6a_release_ab.txt" 2>&1
  echo >> "$O/r6a_release_ab.txt"
done; done
cat "$O/r6a_release_ab.txt"



