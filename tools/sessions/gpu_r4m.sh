#!/usr/bin/env bash
# round 4, session m: Louvain: the RMAT-22 / 24 fixtures on one GPU, the partitioned run (2 / 4 ranks sharing the GPU) against the same fixtures
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "louvain_rmat_golden" 2>&1 | tail -4 | tee "$O/r4m_louvain_golden.log"
timeout 600 python bench_louvain.py --scale 22 --cpu-scale 0 --out "$O/r4m_louvain_s22_sg.json" > /dev/null 2>&1
for w in 2 4; do
  CUGRAPH_AMD_MG_TEST_SINGLE_GPU=1 timeout 900 python bench_louvain.py --gpus $w --scale 22 --repeats 2 --out "$O/r4m_louvain_s22_ranks$w.json" > "$O/r4m_louvain_ranks$w.log" 2>&1; echo "ranks $w rc=$?"; tail -2 "$O/r4m_louvain_ranks$w.log" | cut -c1-300
done
CUGRAPH_AMD_MG_TEST_SINGLE_GPU=1 timeout 900 python bench_louvain.py --gpus 2 --scale 24 --repeats 1 --out "$O/r4m_louvain_s24_ranks2.json" > "$O/r4m_louvain_s24.log" 2>&1; echo "s24 ranks 2 rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4m_louvain_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    print(f.split("/")[-1], "s", d.get("value"), d.get("seconds_all"), "ranks", d.get("n_gpus"), "Q", d.get("modularity"), "sweeps", d.get("sweeps"), "frac", d["roofline"]["frac"], "check", d.get("check"))
PY
