"""Runs the reference's pylibcugraph (its unchanged Cython modules, built by tests/pylibcugraph_run/build.sh) against
libcugraph_c.so on the GPU: the calls and goldens of python/pylibcugraph/pylibcugraph/tests/test_pagerank.py,
test_sssp.py, test_bfs.py (committed as tests/golden/golden.json by tests/golden/make_golden.py).  Prints 'ALL OK'."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "tests" / "pylibcugraph_run" / "_pkg"))
import cupy as cp  # noqa: E402  (the torch-backed stand-in)
import pylibcugraph as plc  # noqa: E402  (the reference's Cython modules)

golden = json.loads((ROOT / "tests" / "golden" / "golden.json").read_text())
handle = plc.ResourceHandle()


def sg_graph(gr, transposed, wdtype=np.float32, vdtype=np.int32):
    props = plc.GraphProperties(is_symmetric=False, is_multigraph=False)
    return plc.SGGraph(handle, props, cp.asarray(gr["src"], dtype=vdtype), cp.asarray(gr["dst"], dtype=vdtype), weight_array=cp.asarray(gr["wgt"], dtype=wdtype),
                       store_transposed=transposed, renumber=True, do_expensive_check=False)


# pagerank.pyx docstring example / C-API goldens
for case in golden["c_api"]["pagerank"]:
    G = sg_graph(case["graph"], case["store_transposed"])
    out = plc.pagerank(handle, G, None, None, None, None, case["alpha"], case["epsilon"], case["max_iterations"], False, fail_on_nonconvergence=False)
    v, pr = out[0].get(), out[1].get()
    got = np.empty(len(pr)); got[v] = pr
    assert np.allclose(got, case["result"], rtol=1e-3), (case["name"], got)
    if not case["converged"]:
        try:
            plc.pagerank(handle, G, None, None, None, None, case["alpha"], case["epsilon"], case["max_iterations"], False)
            raise SystemExit("expected FailedToConvergeError")
        except plc.exceptions.FailedToConvergeError:
            pass
# test_pagerank.py: karate / dolphins goldens
p = golden["pylibcugraph_pagerank"]["params"]
for name in ("karate.csv", "dolphins.csv", "Simple_1", "Simple_2"):
    exp = golden["pylibcugraph_pagerank"][name]
    G = sg_graph(golden["graphs"][name], True)
    v, pr = plc.pagerank(handle, G, None, None, None, None, p["alpha"], p["epsilon"], p["max_iterations"], False)
    got = dict(zip(v.get().tolist(), pr.get().tolist()))
    for vid, val in zip(exp["vertex"], exp["pagerank"]):
        assert abs(got[vid] - val) <= p["rel_tol"] * abs(val) + 5e-7, (name, vid, got[vid], val)
# personalized
for case in golden["c_api"]["personalized_pagerank"]:
    G = sg_graph(case["graph"], case["store_transposed"])
    out = plc.personalized_pagerank(handle, G, None, None, None, None, cp.asarray(case["pers_vertices"], dtype=np.int32),
                                    cp.asarray(case["pers_values"], dtype=np.float32), case["alpha"], case["epsilon"], case["max_iterations"], False,
                                    fail_on_nonconvergence=False)
    got = np.empty(len(case["result"])); got[out[0].get()] = out[1].get()
    assert np.allclose(got, case["result"], rtol=1e-3), case["name"]
# bfs (bfs.pyx calls has_vertex first)
for case in golden["c_api"]["bfs"]:
    G = sg_graph(case["graph"], case["store_transposed"])
    d, pred, v = plc.bfs(handle, G, cp.asarray(case["seeds"], dtype=np.int32), False, case["depth_limit"], True, False)
    order = np.argsort(v.get())
    assert d.get()[order].tolist() == case["distances"] and pred.get()[order].tolist() == case["predecessors"]
# sssp: test_sssp.py goldens (karate / dolphins / Simple_1 / Simple_2)
for name, exp in golden["pylibcugraph_sssp"].items():
    G = sg_graph(golden["graphs"][name], False)
    v, d, pred = plc.sssp(handle, G, exp["start_vertex"], float(np.finfo(np.float32).max), True, False)
    order = np.argsort(v.get())  # (the goldens are in vertex order; this graph is renumbered)
    assert v.get()[order].tolist() == exp["vertex"]
    np.testing.assert_allclose(d.get()[order], np.array(exp["distance"], np.float32), rtol=1e-4)
    if exp["check_predecessor"]:
        assert pred.get()[order].tolist() == exp["predecessor"]
# INT64 columns, python-cugraph's default: the same golden through the int64 path
case = golden["c_api"]["bfs"][0]
G = sg_graph(case["graph"], case["store_transposed"], vdtype=np.int64)
d, pred, v = plc.bfs(handle, G, cp.asarray(case["seeds"], dtype=np.int64), False, case["depth_limit"], True, False)
order = np.argsort(v.get())
big = np.iinfo(np.int64).max
assert d.get()[order].tolist() == [x if x != 2147483647 else big for x in case["distances"]] and v.dtype == np.int64
# degrees, decompress, louvain, generator
case = golden["c_api"]["pagerank"][0]
G = sg_graph(case["graph"], False)
dv, din, dout = plc.degrees(handle, G, None, False)
outd = np.zeros(6, np.int64); outd[dv.get()] = dout.get()
assert outd.tolist() == np.bincount(case["graph"]["src"], minlength=6).tolist()
el = plc.decompress_to_edgelist(handle, G, False)
assert sorted(zip(el[0].get().tolist(), el[1].get().tolist())) == sorted(zip(case["graph"]["src"], case["graph"]["dst"]))
src, dst, wgt, eid, ety = plc.generate_rmat_edgelist(handle, 42, 8, 4096, 0.57, 0.19, 0.19, False, False, True, 0.5, 2.0, np.float32, True, True, 1, 3, False)
assert len(src) == 4096 and int(src.get().max()) < 256 and eid.get().tolist() == list(range(4096))
assert float(wgt.get().min()) >= 0.5 and float(wgt.get().max()) < 2.0 and set(ety.get().tolist()) <= {1, 2, 3}
print("ALL OK")
