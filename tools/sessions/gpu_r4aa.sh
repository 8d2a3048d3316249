#!/usr/bin/env bash
# round 4, session aa: PageRank fixtures (RMAT-22 / 26 against the C oracle's sampled values), pylibcugraph MG runner with the new calls,
# Louvain level sizes (trace), 20 runs of the plain-C conformance binary (the one-off memory fault of session a: does it come back?)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O="$R/gpurun_out"; mkdir -p "$O"; cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_pylibcugraph_on_gpu.py -m gpu -x -q -k "pagerank_rmat_golden or mggraph" --durations=5 2>&1 | tail -12 | tee "$O/r4aa_tests.log"
for sc in 22 26; do CUGRAPH_AMD_LOUVAIN_TRACE=1 timeout 600 python bench_louvain.py --scale $sc --cpu-scale 0 --repeats 1 2>&1 | grep -i "louvain\]\|level" | head -12; done | tee "$O/r4aa_louvain_trace.log"
python - <<'PY' 2>&1 | tail -4 | tee "$R/gpurun_out/r4aa_conformance_loop.log"
import subprocess, sys, tempfile
from pathlib import Path
sys.path.insert(0, "tests")
import test_c_conformance as t
exe = t.build_binary(Path(tempfile.mkdtemp()))
bad = 0
for i in range(25):
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    if r.returncode != 0:
        bad += 1
        print("run", i, "rc", r.returncode, r.stdout[-400:])
print("conformance binary: 25 runs,", bad, "failed")
for name in ("pagerank_test", "bfs_test", "sssp_test"):
    for i in range(5):
        r = subprocess.run([f"tests/c_api/_ref_bin/{name}"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
        if r.returncode != 0 or "FAILED" in r.stdout:
            bad += 1
            print(name, i, "rc", r.returncode, r.stdout[-300:])
print("total failures", bad)
PY
