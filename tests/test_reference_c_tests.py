"""The reference's OWN plain-C API tests (cpp/tests/c_api/*.c), compiled unchanged and in place against include/ and linked to
cugraph_amd/lib/libcugraph_c.so (tests/c_api/build_ref_tests.sh; helper library: tests/c_api/ref_test_shim.c).  The binaries
are built in the build container (where /root/reference exists) and travel to the GPU box with the repo snapshot."""
import os
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BIN = ROOT / "tests" / "c_api" / "_ref_bin"
REF = Path(os.environ.get("CUGRAPH_REFERENCE_DIR", "/root/reference"))
NAMES = ["pagerank_test", "bfs_test", "sssp_test", "louvain_test", "degrees_test", "extract_paths_test", "create_graph_test"]
# the reference's MULTI-GPU tests (started by mpirun there; here tests/c_api/ref_mg_test_shim.c forks the ranks on the library's communicator)
MG_NAMES = ["mg_pagerank_test", "mg_bfs_test", "mg_sssp_test", "mg_louvain_test", "mg_create_graph_test", "mg_degrees_test", "mg_generate_rmat_test"]


def test_reference_c_tests_compile_and_link_unchanged():
    """CPU: every listed reference test compiles against our headers and links with -Wl,--no-undefined."""
    if not (REF / "cpp" / "tests" / "c_api").is_dir():
        pytest.skip("reference tree not present (GPU box): the binaries were built in the build container")
    from cugraph_amd import _capi

    _capi.build()
    out = subprocess.run(["bash", str(ROOT / "tests" / "c_api" / "build_ref_tests.sh")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert out.returncode == 0, out.stdout
    for n in NAMES + MG_NAMES:
        assert (BIN / n).is_file(), f"{n} was not built:\n{out.stdout}"


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_reference_c_test_passes(name):
    """GPU: the reference's test binary reports every test passed (goldens of cpp/tests/c_api/<name>.c through the C ABI)."""
    exe = BIN / name
    if not exe.is_file():
        pytest.skip(f"{exe} missing: built only where the reference tree is available (tests/c_api/build_ref_tests.sh)")
    env = dict(os.environ, LD_LIBRARY_PATH=str(ROOT / "cugraph_amd" / "lib") + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "FAILED" not in out.stdout and "passed" in out.stdout, out.stdout[-4000:]


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 4])
def test_reference_mg_pagerank_c_test_passes_on_the_2d_layout(ranks):
    """Round 6: the reference's mg_pagerank_test.c, unchanged, with the library's multi-GPU PageRank on the reference's own 2-D layout
    (CUGRAPH_AMD_MG_LAYOUT=2d: 1x2 and 2x2 ranks; its personalized cases take the 1-D partition, the plain ones the R x C blocks)."""
    exe = BIN / "mg_pagerank_test"
    if not exe.is_file():
        pytest.skip(f"{exe} missing: built only where the reference tree is available (tests/c_api/build_ref_tests.sh)")
    env = dict(os.environ, LD_LIBRARY_PATH=str(ROOT / "cugraph_amd" / "lib") + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""),
               CUGRAPH_AMD_TEST_RANKS=str(ranks), HSA_ENABLE_IPC_MODE_LEGACY="0", CUGRAPH_AMD_MG_LAYOUT="2d")
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "FAILED" not in out.stdout and "passed" in out.stdout, out.stdout[-4000:]


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])
@pytest.mark.parametrize("name", MG_NAMES)
def test_reference_mg_c_test_passes(name, ranks):
    """GPU: the reference's multi-GPU test binary (cpp/tests/c_api/mg_*_test.c, unchanged) on `ranks` processes sharing the GPU: graph from
    edges on rank 0 (cugraph_graph_create_with_times_mg), collective algorithm calls, every rank checks the vertices it got back against the
    file's goldens, run_mg_test sums the ranks' verdicts."""
    exe = BIN / name
    if not exe.is_file():
        pytest.skip(f"{exe} missing: built only where the reference tree is available (tests/c_api/build_ref_tests.sh)")
    env = dict(os.environ, LD_LIBRARY_PATH=str(ROOT / "cugraph_amd" / "lib") + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""),
               CUGRAPH_AMD_TEST_RANKS=str(ranks), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=env)
    assert out.returncode == 0 and "FAILED" not in out.stdout and "passed" in out.stdout, out.stdout[-4000:]
