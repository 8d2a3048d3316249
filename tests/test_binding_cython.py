"""pylibcugraph's Cython modules on the PageRank / BFS / SSSP path, cythonized IN PLACE from the reference tree (no copies),
compiled against include/ and linked to cugraph_amd/lib/libcugraph_c.so with -Wl,--no-undefined: the unchanged Python layer
finds every header, type and symbol it binds (python/pylibcugraph/pylibcugraph/CMakeLists.txt:78 links cugraph::cugraph_c)."""
import os
import shutil
import subprocess
import sys
import sysconfig
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REF = Path(os.environ.get("CUGRAPH_REFERENCE_DIR", "/root/reference")) / "python" / "pylibcugraph"
MODULES = ["graphs", "utils", "resource_handle", "graph_properties", "pagerank", "personalized_pagerank", "bfs", "sssp", "has_vertex",
           "louvain", "degrees", "decompress_to_edgelist", "random", "generate_rmat_edgelist", "generate_rmat_edgelists"]


def _dlpack_include(tmp: Path) -> Path:
    """dlpack/dlpack.h is a third-party header (dmlc/dlpack, pulled by the reference's get_dlpack.cmake); torch ships a copy."""
    import torch

    src = Path(torch.__file__).parent / "include" / "ATen" / "dlpack.h"
    d = tmp / "inc" / "dlpack"
    d.mkdir(parents=True, exist_ok=True)
    if src.is_file():
        shutil.copy(src, d / "dlpack.h")
    return tmp / "inc"


@pytest.mark.parametrize("module", MODULES)
def test_pyx_binds_unchanged(module, tmp_path):
    if not (REF / "pylibcugraph" / f"{module}.pyx").is_file():
        pytest.skip("reference tree not present")
    pytest.importorskip("Cython")
    from cugraph_amd import _capi

    _capi.build()
    cpp = tmp_path / f"{module}.cpp"
    r = subprocess.run([sys.executable, "-m", "cython", "-3", "--cplus", "-I", str(REF), "-o", str(cpp), str(REF / "pylibcugraph" / f"{module}.pyx")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    pyinc = sysconfig.get_paths()["include"]
    libdir = sysconfig.get_config_var("LIBDIR") or "/usr/lib/x86_64-linux-gnu"
    ver = sysconfig.get_config_var("LDVERSION") or "3.10"
    so = tmp_path / f"{module}.so"
    cmd = ["g++", "-O0", "-w", "-fPIC", "-shared", "-std=c++17", f"-I{ROOT / 'include'}", f"-I{_dlpack_include(tmp_path)}", f"-I{pyinc}", str(cpp), "-o", str(so),
           f"-L{ROOT / 'cugraph_amd' / 'lib'}", "-lcugraph_c", f"-L{libdir}", f"-lpython{ver}", "-Wl,--no-undefined"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and so.is_file(), r.stdout[-4000:]
