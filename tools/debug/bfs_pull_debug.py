"""GPU box: BFS parents, push (atomicMin) against pull (k_bfs_pull_parents), on the parity test's graphs and sources; prints the first mismatches."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2])); sys.path.insert(0, str(Path(__file__).resolve().parents[2] / "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch
import cugraph_amd as cg
from oracle import oracle as orc
from conftest import rmat_graph
from test_gpu_parity import make_graph, by_vertex, T, bfs_expected_parents

h = cg.ResourceHandle()
for scale, transposed in ((16, True), (18, False)):
    s, d = rmat_graph(orc, scale)
    nv = 1 << scale
    g = make_graph(cg, h, s, d, None, transposed=transposed, renumber=True, vertices=np.arange(nv))
    off, idx, _ = orc.coo_to_cs(nv, s, d)
    indeg = np.bincount(d, minlength=nv)
    for src in [int(x) for x in np.nonzero(np.diff(off) > 0)[0][[0, 7, 100]]]:
        res = {}
        for mode in ("0", "1"):
            os.environ["CUGRAPH_AMD_BFS_PULL_PARENTS"] = mode
            os.environ["CUGRAPH_AMD_BFS_TRACE"] = "1" if mode == "1" else ""
            if mode == "0":
                os.environ.pop("CUGRAPH_AMD_BFS_TRACE", None)
            dist, pred, v = cg.bfs(h, g, T([src], np.int32), False, 0, True, False)
            verts = v
            res[mode] = by_vertex(v, dist, pred)
        od, _ = orc.bfs(nv, off, idx, [src])
        exp = bfs_expected_parents(s, d, od, verts)
        vv = verts.cpu().numpy().astype(np.int64); int_of = np.empty(nv, np.int64); int_of[vv] = np.arange(nv)
        for mode in ("0", "1"):
            bad = np.nonzero(res[mode][1] != exp)[0]
            print(f"scale {scale} transposed {transposed} src {src} pull={mode}: dist ok {np.array_equal(res[mode][0], od)}, {bad.size} parent mismatches", flush=True)
            for x in bad[:6]:
                p, e = int(res[mode][1][x]), int(exp[x])
                print(f"   v ext {x} int {int_of[x]} depth {od[x]} in-degree {indeg[x]}: got parent ext {p} (int {int_of[p] if p >= 0 else -1}, depth {od[p] if p >= 0 else None}) "
                      f"expected ext {e} (int {int_of[e] if e >= 0 else -1}, depth {od[e] if e >= 0 else None})", flush=True)
