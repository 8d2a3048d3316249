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
