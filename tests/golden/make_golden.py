#!/usr/bin/env python3
"""Generates tests/golden/golden.json from the reference's own tests (run in the build container, where
/root/reference exists; the GPU box only sees the committed JSON).

Sources of the vectors:
  * python/pylibcugraph/pylibcugraph/tests/test_pagerank.py:14-147  (_test_data: karate, dolphins, Simple_1/2)
    and tests/test_sssp.py:14-298 -- imported here with a stub `cupy` module (cupy.asarray -> numpy),
    so the numbers are read from the reference files, not retyped;
  * python/pylibcugraph/pylibcugraph/tests/conftest.py:44-57 (Simple_1 / Simple_2 edge lists), same way;
  * datasets/karate.csv, datasets/dolphins.csv (space separated `src dst weight`), embedded as edge lists;
  * cpp/tests/c_api/pagerank_test.c:385-544, bfs_test.c:160-210, sssp_test.c:167-223: plain-C literals,
    transcribed below with their line numbers.
"""
import importlib.util
import json
import sys
import types
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent / "golden.json"


def load_with_stub_cupy(path, name):
    cp = types.ModuleType("cupy")
    cp.asarray = lambda x, dtype=None: np.asarray(list(x) if isinstance(x, range) else x, dtype=dtype)
    cp.ndarray = np.ndarray
    sys.modules["cupy"] = cp
    pt = types.ModuleType("pytest")
    pt.fixture = lambda *a, **k: (lambda f: f)
    pt.param = lambda *a, **k: a
    pt.approx = lambda *a, **k: a
    pt.mark = types.SimpleNamespace(skip=lambda *a, **k: (lambda f: f), parametrize=lambda *a, **k: (lambda f: f))
    saved = sys.modules.get("pytest")
    sys.modules["pytest"] = pt
    try:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        if saved is not None:
            sys.modules["pytest"] = saved
        else:
            del sys.modules["pytest"]
        del sys.modules["cupy"]
    return mod


def read_csv(path):
    a = np.loadtxt(path)
    return a[:, 0].astype(int).tolist(), a[:, 1].astype(int).tolist(), a[:, 2].astype(float).tolist()


def main():
    tdir = REF / "python/pylibcugraph/pylibcugraph/tests"
    pr = load_with_stub_cupy(tdir / "test_pagerank.py", "ref_test_pagerank")
    ss = load_with_stub_cupy(tdir / "test_sssp.py", "ref_test_sssp")
    g = {"graphs": {}, "pylibcugraph_pagerank": {}, "pylibcugraph_sssp": {}, "c_api": {}}
    for name in ("karate.csv", "dolphins.csv"):
        s, d, w = read_csv(REF / "datasets" / name)
        g["graphs"][name] = {"src": s, "dst": d, "wgt": w}
    # conftest.py:44-57
    g["graphs"]["Simple_1"] = {"src": [0, 1, 2], "dst": [1, 2, 3], "wgt": [1.0, 1.0, 1.0]}
    g["graphs"]["Simple_2"] = {"src": [0, 1, 1, 2, 2, 2, 3, 4], "dst": [1, 3, 4, 0, 1, 3, 5, 5],
                               "wgt": [0.1, 2.1, 1.1, 5.1, 3.1, 4.1, 7.2, 3.2]}
    # test_pagerank.py: alpha / epsilon / max_iterations at :14-16, tolerance rel 1e-4 at the tail
    g["pylibcugraph_pagerank"]["params"] = {"alpha": pr._alpha, "epsilon": pr._epsilon, "max_iterations": pr._max_iterations,
                                            "rel_tol": 1e-4}
    for name, (verts, vals) in pr._test_data.items():
        g["pylibcugraph_pagerank"][name] = {"vertex": np.asarray(verts).tolist(), "pagerank": [float(x) for x in np.asarray(vals, dtype=np.float64)]}
    for name, d in ss._test_data.items():
        g["pylibcugraph_sssp"][name] = {
            "start_vertex": int(d["start_vertex"]), "vertex": np.asarray(d["vertex"]).tolist(),
            "distance": [float(x) for x in np.asarray(d["distance"], dtype=np.float64)],
            "predecessor": np.asarray(d["predecessor"]).tolist(),
            # test_sssp.py tail: predecessors are compared only for Simple_1 / Simple_2
            "check_predecessor": name in ("Simple_1", "Simple_2"),
        }
    FLT_MAX = float(np.finfo(np.float32).max)
    DBL_MAX = float(np.finfo(np.float64).max)
    six = {"src": [0, 1, 1, 2, 2, 2, 3, 4], "dst": [1, 3, 4, 0, 1, 3, 5, 5], "wgt": [0.1, 2.1, 1.1, 5.1, 3.1, 4.1, 7.2, 3.2]}
    chain = {"src": [0, 1, 2], "dst": [1, 2, 3], "wgt": [1.0, 1.0, 1.0]}
    g["c_api"] = {
        "tolerance": 0.001,  # c_test_utils.h nearlyEqual(..., 0.001)
        "pagerank": [
            # pagerank_test.c:385-402 test_pagerank / :404-423 test_pagerank_with_transpose
            {"name": "test_pagerank", "graph": six, "store_transposed": True, "alpha": 0.95, "epsilon": 0.0001, "max_iterations": 20,
             "result": [0.0915528, 0.168382, 0.0656831, 0.191468, 0.120677, 0.362237], "converged": True},
            {"name": "test_pagerank_with_transpose", "graph": six, "store_transposed": False, "alpha": 0.95, "epsilon": 0.0001,
             "max_iterations": 20, "result": [0.0915528, 0.168382, 0.0656831, 0.191468, 0.120677, 0.362237], "converged": True},
            # :425-442 test_pagerank_4 / :444-461 test_pagerank_4_with_transpose
            {"name": "test_pagerank_4", "graph": chain, "store_transposed": False, "alpha": 0.85, "epsilon": 1.0e-6, "max_iterations": 500,
             "result": [0.11615584790706635, 0.21488840878009796, 0.29881080985069275, 0.37014490365982056], "converged": True},
            {"name": "test_pagerank_4_with_transpose", "graph": chain, "store_transposed": True, "alpha": 0.85, "epsilon": 1.0e-6,
             "max_iterations": 500, "result": [0.11615584790706635, 0.21488840878009796, 0.29881080985069275, 0.37014490365982056],
             "converged": True},
            # :463-480 test_pagerank_non_convergence
            {"name": "test_pagerank_non_convergence", "graph": six, "store_transposed": True, "alpha": 0.95, "epsilon": 0.0001,
             "max_iterations": 2, "result": [0.0776471, 0.167637, 0.0639699, 0.220202, 0.140046, 0.330498], "converged": False},
        ],
        "personalized_pagerank": [
            # :482-512 test_personalized_pagerank / :514-544 test_personalized_pagerank_non_convergence
            {"name": "test_personalized_pagerank", "graph": chain, "store_transposed": False, "alpha": 0.85, "epsilon": 1.0e-6,
             "max_iterations": 500, "pers_vertices": [0, 1, 2, 3], "pers_values": [0.1, 0.2, 0.3, 0.4],
             "result": [0.0559233, 0.159381, 0.303244, 0.481451], "converged": True},
            {"name": "test_personalized_pagerank_non_convergence", "graph": chain, "store_transposed": False, "alpha": 0.85,
             "epsilon": 1.0e-6, "max_iterations": 1, "pers_vertices": [0, 1, 2, 3], "pers_values": [0.1, 0.2, 0.3, 0.4],
             "result": [0.03625, 0.285, 0.32125, 0.3575], "converged": False},
        ],
        "bfs": [
            # bfs_test.c:160-184 test_bfs / :186-210 test_bfs_with_transpose
            {"name": "test_bfs", "graph": six, "store_transposed": False, "seeds": [0], "depth_limit": 10,
             "distances": [0, 1, 2147483647, 2, 2, 3], "predecessors": [-1, 0, -1, 1, 1, 3]},
            {"name": "test_bfs_with_transpose", "graph": six, "store_transposed": True, "seeds": [0], "depth_limit": 10,
             "distances": [0, 1, 2147483647, 2, 2, 3], "predecessors": [-1, 0, -1, 1, 1, 3]},
        ],
        "sssp": [
            # sssp_test.c:167-189 test_sssp / :191-206 test_sssp_with_transpose / :208-223 ..._double
            {"name": "test_sssp", "graph": six, "dtype": "float32", "store_transposed": False, "source": 0,
             "distances": [0.0, 0.1, FLT_MAX, 2.2, 1.2, 4.4], "predecessors": [-1, 0, -1, 1, 1, 4]},
            {"name": "test_sssp_with_transpose", "graph": six, "dtype": "float32", "store_transposed": True, "source": 0,
             "distances": [0.0, 0.1, FLT_MAX, 2.2, 1.2, 4.4], "predecessors": [-1, 0, -1, 1, 1, 4]},
            {"name": "test_sssp_with_transpose_double", "graph": six, "dtype": "float64", "store_transposed": True, "source": 0,
             "distances": [0.0, 0.1, DBL_MAX, 2.2, 1.2, 4.4], "predecessors": [-1, 0, -1, 1, 1, 4]},
        ],
    }
    OUT.write_text(json.dumps(g, indent=1))
    print("wrote", OUT, OUT.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
