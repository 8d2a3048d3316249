"""One rank of a multi-GPU job driven through the reference's pylibcugraph -- its UNCHANGED Cython modules (tests/pylibcugraph_run/build.sh):
plc.ResourceHandle(handle=<address of the communicator>) where the reference passes the address of a raft::handle_t
(resource_handle.pyx:47-66), plc.MGGraph with this rank's slice of the edge list (graphs.pyx:357-700), then plc.pagerank / plc.bfs /
plc.sssp / plc.personalized_pagerank / plc.degrees: the karate and C-API goldens of the single-GPU runner (run_goldens.py) must come back,
each vertex from exactly one rank.
usage: run_mg.py <session> <rank> <size> <outdir>; prints 'RANK OK'."""
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "tests" / "pylibcugraph_run" / "_pkg"))
import cupy as cp  # noqa: E402  (the torch-backed stand-in)
import pylibcugraph as plc  # noqa: E402  (the reference's Cython modules)

session, rank, size, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), Path(sys.argv[4])
lib = C.CDLL(str(ROOT / "cugraph_amd" / "lib" / "libcugraph_c.so"))  # the library the Cython modules are linked to
lib.cugraph_amd_comm_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
lib.cugraph_error_message.restype = C.c_char_p
lib.cugraph_error_message.argtypes = [C.c_void_p]
lib.cugraph_amd_comm_free.argtypes = [C.c_void_p]
comm, err = C.c_void_p(), C.c_void_p()
rc = lib.cugraph_amd_comm_create(session.encode(), rank, size, C.byref(comm), C.byref(err))
assert rc == 0, lib.cugraph_error_message(err)
handle = plc.ResourceHandle(handle=comm.value)
golden = json.loads((ROOT / "tests" / "golden" / "golden.json").read_text())
out = {}


def mg_graph(gr, transposed):
    src, dst, wgt = (np.asarray(gr[k]) for k in ("src", "dst", "wgt"))
    mine = np.arange(src.size) % size == rank  # this rank's slice
    props = plc.GraphProperties(is_symmetric=False, is_multigraph=False)
    return plc.MGGraph(handle, props, [cp.asarray(src[mine], dtype=np.int32)], [cp.asarray(dst[mine], dtype=np.int32)],
                       weight_array=[cp.asarray(wgt[mine], dtype=np.float32)], store_transposed=transposed, num_arrays=1, do_expensive_check=False)


# PageRank: the C-API golden (pagerank_test.c) and karate (test_pagerank.py)
case = golden["c_api"]["pagerank"][0]
G = mg_graph(case["graph"], True)
v, pr = plc.pagerank(handle, G, None, None, None, None, case["alpha"], case["epsilon"], case["max_iterations"], False, fail_on_nonconvergence=False)[:2]
out["capi_pagerank"] = {"v": v.get().tolist(), "x": pr.get().tolist(), "want": case["result"]}
p = golden["pylibcugraph_pagerank"]["params"]
G = mg_graph(golden["graphs"]["karate.csv"], True)
v, pr = plc.pagerank(handle, G, None, None, None, None, p["alpha"], p["epsilon"], p["max_iterations"], False)
exp = golden["pylibcugraph_pagerank"]["karate.csv"]
out["karate_pagerank"] = {"v": v.get().tolist(), "x": pr.get().tolist(), "want_v": exp["vertex"], "want": exp["pagerank"], "rel_tol": p["rel_tol"]}
# BFS (bfs.pyx calls has_vertex first) and SSSP on the C-API graph
case = golden["c_api"]["bfs"][0]
G = mg_graph(case["graph"], False)
d, pred, v = plc.bfs(handle, G, cp.asarray(case["seeds"], dtype=np.int32), False, case["depth_limit"], True, False)
out["bfs"] = {"v": v.get().tolist(), "d": d.get().tolist(), "p": pred.get().tolist(), "want_d": case["distances"], "want_p": case["predecessors"]}
exp = golden["pylibcugraph_sssp"]["karate.csv"]
G = mg_graph(golden["graphs"]["karate.csv"], False)
v, d, pred = plc.sssp(handle, G, exp["start_vertex"], float(np.finfo(np.float32).max), True, False)
out["karate_sssp"] = {"v": v.get().tolist(), "d": d.get().tolist(), "want_v": exp["vertex"], "want_d": exp["distance"]}
# personalized PageRank (pagerank_test.c: test_personalized_pagerank): rank 0 names all personalization vertices, whoever owns them
case = golden["c_api"]["personalized_pagerank"][0]
G = mg_graph(case["graph"], True)
npers = len(case["pers_vertices"]) if rank == 0 else 0
v, pr = plc.personalized_pagerank(handle, G, None, None, None, None, cp.asarray(case["pers_vertices"][:npers], dtype=np.int32),
                                  cp.asarray(case["pers_values"][:npers], dtype=np.float32), case["alpha"], case["epsilon"], case["max_iterations"], False,
                                  fail_on_nonconvergence=False)[:2]
out["capi_ppr"] = {"v": v.get().tolist(), "x": pr.get().tolist(), "want": case["result"]}
# SSSP with FLOAT64 weights (sssp_test.c: test_sssp_with_transpose_double)
case = [c for c in golden["c_api"]["sssp"] if c["dtype"] == "float64"][0]
gr = case["graph"]
src, dst, wgt = (np.asarray(gr[k]) for k in ("src", "dst", "wgt"))
mine = np.arange(src.size) % size == rank
G = plc.MGGraph(handle, plc.GraphProperties(is_symmetric=False, is_multigraph=False), [cp.asarray(src[mine], dtype=np.int32)], [cp.asarray(dst[mine], dtype=np.int32)],
                weight_array=[cp.asarray(wgt[mine], dtype=np.float64)], store_transposed=False, num_arrays=1, do_expensive_check=False)
v, d, pred = plc.sssp(handle, G, case["source"], float(np.finfo(np.float64).max), True, False)
out["sssp_f64"] = {"v": v.get().tolist(), "d": d.get().tolist(), "p": pred.get().tolist(), "want_d": case["distances"], "want_p": case["predecessors"]}
# degrees of the same graph: every rank's share
v, din, dout = plc.degrees(handle, G, None, False)
out["degrees"] = {"v": v.get().tolist(), "in": din.get().tolist(), "out": dout.get().tolist(), "src": src.tolist(), "dst": dst.tolist()}
(outdir / f"plc_rank{rank}.json").write_text(json.dumps(out))
del G, handle
lib.cugraph_amd_comm_free(comm)
print("RANK OK")
