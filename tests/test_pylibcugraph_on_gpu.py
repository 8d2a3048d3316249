"""The reference's pylibcugraph -- its UNCHANGED Cython modules, cythonized in place from the reference tree and linked to
libcugraph_c.so (tests/pylibcugraph_run/build.sh, built where the reference tree exists) -- driven on the GPU through its own
public API with the goldens of its own tests.  cupy, which those modules import for result arrays, is replaced by a
torch-backed stand-in (tests/pylibcugraph_run/standins/cupy); everything between the Python call and the kernels is the
reference's binding code and this library."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "tests" / "pylibcugraph_run" / "_pkg" / "pylibcugraph"
REF = Path(os.environ.get("CUGRAPH_REFERENCE_DIR", "/root/reference")) / "python" / "pylibcugraph"


def test_pylibcugraph_package_builds_and_imports():
    """CPU: the package builds from the reference tree and imports (every Cython module resolves its C symbols at load time)."""
    if not (REF / "pylibcugraph" / "graphs.pyx").is_file():
        pytest.skip("reference tree not present (GPU box): the package was built in the build container")
    pytest.importorskip("Cython")
    from cugraph_amd import _capi

    _capi.build()
    out = subprocess.run(["bash", str(ROOT / "tests" / "pylibcugraph_run" / "build.sh")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert out.returncode == 0, out.stdout[-3000:]
    env = dict(os.environ, PYTHONPATH=str(PKG.parent))
    r = subprocess.run([sys.executable, "-c", "import pylibcugraph as p; print(p.SGGraph, p.pagerank, p.bfs, p.sssp)"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, env=env, cwd="/tmp")
    assert r.returncode == 0, r.stdout[-3000:]


@pytest.mark.gpu
def test_pylibcugraph_goldens_through_the_reference_binding():
    if not any(PKG.glob("pagerank*.so")):
        pytest.skip("tests/pylibcugraph_run/_pkg missing: built only where the reference tree is available")
    r = subprocess.run([sys.executable, str(ROOT / "tests" / "pylibcugraph_run" / "run_goldens.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=600, cwd="/tmp")
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-4000:]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_pylibcugraph_mggraph_on_ranks_sharing_one_gpu(world, tmp_path):
    """The reference's MGGraph + pagerank / bfs / sssp -- unchanged Cython modules -- on several ranks of the library's communicator
    (ResourceHandle(handle=<communicator>), graphs.pyx:357): the single-GPU goldens come back, every vertex from exactly one rank."""
    import json
    import uuid

    import numpy as np

    if not any(PKG.glob("pagerank*.so")):
        pytest.skip("tests/pylibcugraph_run/_pkg missing: built only where the reference tree is available")
    session = f"plc{uuid.uuid4().hex[:10]}"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CUGRAPH_AMD_COMM_TIMEOUT_S="60")
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "pylibcugraph_run" / "run_mg.py"), session, str(r), str(world), str(tmp_path)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True, env=env, cwd="/tmp") for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "RANK OK" in o, o[-4000:]
    res = [json.loads((tmp_path / f"plc_rank{r}.json").read_text()) for r in range(world)]

    def merged(key, col):
        got = {}
        for r in res:
            for v, x in zip(r[key]["v"], r[key][col]):
                assert v not in got, "a vertex came back from two ranks"
                got[v] = x
        return got

    pr = merged("capi_pagerank", "x")
    want = res[0]["capi_pagerank"]["want"]
    assert sorted(pr) == list(range(len(want))) and np.allclose([pr[v] for v in range(len(want))], want, rtol=1e-3)
    k = res[0]["karate_pagerank"]
    pr = merged("karate_pagerank", "x")
    for vid, val in zip(k["want_v"], k["want"]):
        assert abs(pr[vid] - val) <= k["rel_tol"] * abs(val) + 5e-7
    d, p = merged("bfs", "d"), merged("bfs", "p")
    assert [d[v] for v in sorted(d)] == res[0]["bfs"]["want_d"] and [p[v] for v in sorted(p)] == res[0]["bfs"]["want_p"]
    d = merged("karate_sssp", "d")
    k = res[0]["karate_sssp"]
    np.testing.assert_allclose([d[v] for v in k["want_v"]], np.array(k["want_d"], np.float32), rtol=1e-4)
    pr = merged("capi_ppr", "x")
    want = res[0]["capi_ppr"]["want"]
    assert sorted(pr) == list(range(len(want))) and np.allclose([pr[v] for v in range(len(want))], want, rtol=1e-3)
    d, p = merged("sssp_f64", "d"), merged("sssp_f64", "p")
    k = res[0]["sssp_f64"]
    np.testing.assert_allclose([d[v] for v in sorted(d)], k["want_d"], rtol=1e-12)
    assert [p[v] for v in sorted(p)] == k["want_p"]
    din, dout = merged("degrees", "in"), merged("degrees", "out")
    k = res[0]["degrees"]
    nv = max(max(k["src"]), max(k["dst"])) + 1
    assert [din[v] for v in range(nv)] == np.bincount(k["dst"], minlength=nv).tolist() and [dout[v] for v in range(nv)] == np.bincount(k["src"], minlength=nv).tolist()
