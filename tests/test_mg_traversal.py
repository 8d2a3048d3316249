"""Multi-process BFS / SSSP (cugraph_amd/mg_traversal.py): world_size 2/3 gloo on CPU with the numpy engine (partitioning +
collectives), and -- on the GPU box -- the HIP engine driven by 1, 2 and 4 ranks sharing cuda:0.  Distances must equal the
oracle's bit for bit; parents are checked the way the reference's tests do (bfs_test.cpp:217-233, sssp_test.cpp) and
against the minimum-external-id rule this path promises."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import rmat_graph

ROOT = Path(__file__).resolve().parent.parent
INT32_MAX = 2**31 - 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_world(engine, algo, world, scale, tmp_path, sources, limit=-1.0, direction="", stats=None, extra_env=None):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", CUGRAPH_AMD_MG_BFS=direction, **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "mg_traversal_worker.py"), engine, algo, str(scale), str(tmp_path),
                                       ",".join(str(x) for x in sources), str(limit)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    nv = 1 << scale
    seen = np.zeros(nv, np.int32)
    dist = np.zeros(nv, res[0]["d"].dtype)
    pred = np.zeros(nv, np.int32)
    for r in res:
        seen[r["v"]] += 1
        dist[r["v"]] = r["d"]
        pred[r["v"]] = r["p"]
    assert (seen == 1).all(), "every vertex must be owned by exactly one rank"
    if stats is not None:
        stats.update(levels=int(res[0]["levels"]), bottom_up_levels=int(res[0]["bottom_up_levels"]))
    return dist, pred


def graph(orc, scale, weighted=False):
    s, d = rmat_graph(orc, scale)
    nv = 1 << scale
    w = np.random.default_rng(1).integers(1, 256, size=s.size).astype(np.float32) if weighted else None
    off, idx, ww = orc.coo_to_cs(nv, s, d, w)  # CSR: rows are sources
    return nv, s, d, w, off, idx, ww


def min_ext_parent(nv, s, d, ok):
    """minimum source id over the edges selected by `ok` per destination (-1: none)"""
    best = np.full(nv, INT32_MAX, np.int64)
    np.minimum.at(best, d[ok], s[ok])
    return np.where(best == INT32_MAX, -1, best).astype(np.int32)


def check_bfs(orc, scale, sources, dist, pred, depth_limit=INT32_MAX):
    nv, s, d, _, off, idx, _ = graph(orc, scale)
    od, _ = orc.bfs(nv, off, idx, np.asarray(sources, np.int32), depth_limit)
    assert np.array_equal(dist, od)
    ok = (od[s] != INT32_MAX) & (od[d] != INT32_MAX) & (od[s].astype(np.int64) + 1 == od[d])
    want = min_ext_parent(nv, s, d, ok)
    want[np.asarray(sources)] = -1
    assert np.array_equal(pred, want)


def check_sssp(orc, scale, source, dist, pred, cutoff=np.inf):
    nv, s, d, w, off, idx, ww = graph(orc, scale, weighted=True)
    od, _ = orc.sssp(nv, off, idx, ww, source, cutoff)
    od = od.astype(np.float32)
    assert np.array_equal(dist.view(np.uint32), od.view(np.uint32))
    fmax = np.finfo(np.float32).max
    ok = (od[s] != fmax) & (od[d] != fmax) & ((od[s] + w).astype(np.float32) == od[d])
    want = min_ext_parent(nv, s, d, ok)
    want[source] = -1
    assert np.array_equal(pred, want)


def pick_sources(orc, scale, n):
    s, _ = rmat_graph(orc, scale)
    deg = np.bincount(s, minlength=1 << scale)
    cand = np.flatnonzero(deg > 0)
    return [int(x) for x in cand[np.random.default_rng(3).choice(cand.size, n, replace=False)]]


@pytest.mark.parametrize("world", [2, 3])
def test_mg_bfs_gloo_cpu(orc, tmp_path, world):
    scale = 10
    src = pick_sources(orc, scale, 2)
    dist, pred = run_world("numpy", "bfs", world, scale, tmp_path, src)
    check_bfs(orc, scale, src, dist, pred)


@pytest.mark.parametrize("world,direction", [(2, "bottomup"), (3, "bottomup"), (2, "topdown"), (4, "")])
def test_mg_bfs_gloo_cpu_directions(orc, tmp_path, world, direction):
    """Direction-optimising partitioned BFS: every level bottom-up (each unvisited owned row scans its in-edges against the all-gathered
    frontier bitmap; no candidate exchange), every level top-down, and the heuristic's mix -- same distances, same minimum-external-id
    parents."""
    scale = 12
    src = pick_sources(orc, scale, 2)
    st = {}
    dist, pred = run_world("numpy", "bfs", world, scale, tmp_path, src, direction=direction, stats=st)
    check_bfs(orc, scale, src, dist, pred)
    if direction == "bottomup":
        assert st["bottom_up_levels"] == st["levels"]
    if direction == "topdown":
        assert st["bottom_up_levels"] == 0
    if direction == "":
        assert 0 < st["bottom_up_levels"] < st["levels"]  # RMAT-12: the two widest levels run bottom-up


@pytest.mark.parametrize("direction", ["bottomup", ""])
def test_mg_bfs_gloo_cpu_tiny_graph_many_ranks(orc, tmp_path, direction):
    """8 vertices over 5 ranks (partitions of one or two rows, padded to 64): the in-edge copy, the external-id table and the bitmaps hold."""
    src = pick_sources(orc, 3, 1)
    dist, pred = run_world("numpy", "bfs", 5, 3, tmp_path, src, direction=direction)
    check_bfs(orc, 3, src, dist, pred)


def test_mg_bfs_gloo_cpu_depth_limit(orc, tmp_path):
    scale = 10
    src = pick_sources(orc, scale, 1)
    dist, pred = run_world("numpy", "bfs", 2, scale, tmp_path, src, limit=2)
    check_bfs(orc, scale, src, dist, pred, depth_limit=2)


@pytest.mark.parametrize("world", [2, 3])
def test_mg_sssp_gloo_cpu(orc, tmp_path, world):
    scale = 9
    src = pick_sources(orc, scale, 1)
    dist, pred = run_world("numpy", "sssp", world, scale, tmp_path, src)
    check_sssp(orc, scale, src[0], dist, pred)


def test_mg_sssp_gloo_cpu_cutoff(orc, tmp_path):
    scale = 9
    src = pick_sources(orc, scale, 1)
    dist, pred = run_world("numpy", "sssp", 2, scale, tmp_path, src, limit=300.0)
    check_sssp(orc, scale, src[0], dist, pred, cutoff=300.0)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 4])
def test_mg_bfs_hip_engine(orc, tmp_path, world):
    scale = 14
    src = pick_sources(orc, scale, 3)
    dist, pred = run_world("hip", "bfs", world, scale, tmp_path, src)
    check_bfs(orc, scale, src, dist, pred)


@pytest.mark.gpu
@pytest.mark.parametrize("world,direction", [(1, "bottomup"), (2, "bottomup"), (4, "bottomup"), (2, "topdown"), (2, ""), (4, "")])
def test_mg_bfs_hip_engine_directions(orc, tmp_path, world, direction):
    """the bottom-up level of the partitioned HIP engine (k_bfs_bottom_up on the local in-edges, frontier = the all-gathered bitmap)"""
    scale = 14
    src = pick_sources(orc, scale, 2)
    st = {}
    dist, pred = run_world("hip", "bfs", world, scale, tmp_path, src, direction=direction, stats=st)
    check_bfs(orc, scale, src, dist, pred)
    if direction == "bottomup":
        assert st["bottom_up_levels"] == st["levels"]
    if direction == "":
        assert 0 < st["bottom_up_levels"] < st["levels"]


@pytest.mark.gpu
def test_mg_bfs_hip_engine_depth_limit(orc, tmp_path):
    scale = 12
    src = pick_sources(orc, scale, 1)
    dist, pred = run_world("hip", "bfs", 2, scale, tmp_path, src, limit=3)
    check_bfs(orc, scale, src, dist, pred, depth_limit=3)


@pytest.mark.gpu
@pytest.mark.parametrize("world,in_place", [(1, "1"), (2, "1"), (4, "1"), (3, "1"), (1, "0"), (2, "0")])
def test_mg_sssp_hip_engine(orc, tmp_path, world, in_place):
    """in_place = 1 (default): candidates whose destination the expanding rank owns are relaxed in place during expand (with one rank:
    all of them -- nothing is exchanged); 0: every candidate goes through the sender-side table, the exchange and apply.  Same fixed point:
    distances bit-identical to Dijkstra, minimum-external-id parents among the tight in-edges."""
    scale = 13
    src = pick_sources(orc, scale, 1)
    dist, pred = run_world("hip", "sssp", world, scale, tmp_path, src, extra_env={"CUGRAPH_AMD_MG_SSSP_INPLACE": in_place})
    check_sssp(orc, scale, src[0], dist, pred)


@pytest.mark.gpu
def test_mg_sssp_hip_engine_cutoff(orc, tmp_path):
    scale = 12
    src = pick_sources(orc, scale, 1)
    dist, pred = run_world("hip", "sssp", 2, scale, tmp_path, src, limit=400.0)
    check_sssp(orc, scale, src[0], dist, pred, cutoff=400.0)
