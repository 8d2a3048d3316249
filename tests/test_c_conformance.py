"""A plain-C caller (tests/c_api/conformance.c, gcc -std=c99 -Wall -Wextra -Werror) compiles against include/ and links
libcugraph_c.so: the headers are a usable C ABI without any HIP / C++ / torch type.  On the GPU box the binary runs the
reference's C-test golden vectors (PageRank, BFS, SSSP on the 6-vertex graph, all four store_transposed x renumber
variants) through that ABI."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def build_binary(tmp_path, source="conformance.c"):
    from cugraph_amd import _capi

    if not _capi.LIB_PATH.exists():
        _capi.build()
    exe = tmp_path / source[:-2]
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", f"-I{ROOT / 'include'}", str(ROOT / "tests" / "c_api" / source), "-o", str(exe),
           f"-L{_capi.LIB_DIR}", "-lcugraph_c", "-lm", f"-Wl,-rpath,{_capi.LIB_DIR}"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout
    return exe


def test_plain_c_caller_compiles_and_links(tmp_path):
    assert build_binary(tmp_path).exists()


@pytest.mark.gpu
def test_plain_c_caller_reproduces_the_reference_goldens(tmp_path):
    exe = build_binary(tmp_path)
    res = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert res.returncode == 0, res.stdout
    assert "c_api conformance: ok" in res.stdout


def test_plain_c_multi_rank_caller_compiles_and_links(tmp_path):
    assert build_binary(tmp_path, "mg_two_ranks.c").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])
def test_plain_c_caller_drives_ranks_through_the_mg_entry_points(tmp_path, ranks):
    """tests/c_api/mg_two_ranks.c: forked processes, each a rank on the library's communicator, cugraph_graph_create_mg + cugraph_pagerank /
    cugraph_bfs / cugraph_sssp against the goldens of the reference's (single- and multi-GPU) C tests"""
    import os

    exe = build_binary(tmp_path, "mg_two_ranks.c")
    res = subprocess.run([str(exe), str(ranks)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CUGRAPH_AMD_COMM_TIMEOUT_S="60"))
    assert res.returncode == 0 and "c_api multi-rank: ok" in res.stdout, res.stdout[-3000:]
