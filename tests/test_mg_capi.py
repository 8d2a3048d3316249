"""Multi-GPU behind the reference's own entry points (cugraph_create_resource_handle on a communicator -> cugraph_graph_create_mg ->
cugraph_pagerank / cugraph_bfs / cugraph_sssp / cugraph_louvain; SURVEY.md section 8e, c_api/graph_mg.cpp:140,326,
c_api/resource_handle.cpp:11-39): 2 / 3 / 4 processes share cuda:0 of the GPU box, the library's communicator (HIP IPC windows + peer
writes) carries the exchanges -- the same code path as one process per GPU on an xGMI node."""
import json
import os
import subprocess
import sys
import uuid
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def run_ranks(what, world, tmp_path, *args, timeout=600, env_extra=None):
    session = f"t{uuid.uuid4().hex[:12]}"
    procs = []
    for rank in range(world):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CUGRAPH_AMD_COMM_TIMEOUT_S="60", **(env_extra or {}))
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "ipc_worker.py"), what, session, str(rank), str(world), str(tmp_path)] + [str(a) for a in args],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=timeout)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}:\n{o[-4000:]}"
    run_ranks.last_outputs = outs  # (what the ranks printed: debug lines of the library)
    return [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(world)]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 4])
def test_comm_selftest(world, tmp_path):
    """every primitive of the communicator against closed forms (all-gather, all-to-all-v, integer / double all-reduce in rank order),
    several ranks sharing one GPU"""
    res = run_ranks("selftest", world, tmp_path, 1 << 16, 20)
    assert len(res) == world
    for r in res:
        assert r["barrier_us"] > 0 and r["push_gbps"] > 0


@pytest.mark.parametrize("world", [2, 5])
def test_comm_host_bootstrap(world):
    """the shared-memory bootstrap of the communicator (session attach, sense-reversing barrier, host all-gather) with real processes;
    no GPU involved"""
    session = f"h{uuid.uuid4().hex[:12]}"
    code = ("import ctypes as C, sys; from cugraph_amd import _capi as capi; l = capi.lib(); e = C.c_void_p(); "
            "rc = l.cugraph_amd_comm_host_selftest(sys.argv[1].encode(), int(sys.argv[2]), int(sys.argv[3]), 200, C.byref(e)); "
            "print(l.cugraph_error_message(e) if rc else 'ok'); sys.exit(rc)")
    procs = [subprocess.Popen([sys.executable, "-c", code, session, str(r), str(world)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-2000:]
    assert not Path("/dev/shm/cga_" + session).exists()


@pytest.mark.parametrize("creator", ["dead", "alive", "aborted", "foreign"])
def test_comm_host_bootstrap_survives_a_stale_segment(creator):
    """A crashed job left a segment of the same session name behind (ready, one arrival short of a full barrier), and this time the other ranks
    start BEFORE rank 0 (advisor finding, round 4).  They must not settle on the old segment: its creator is gone (dead pid), or -- the pid
    has been recycled and looks alive -- rank 0 re-creates the session under their feet and they move over when the name no longer leads to
    the file they mapped.  No GPU involved."""
    import struct
    import time

    world = 3
    session = f"s{uuid.uuid4().hex[:12]}"
    path = Path("/dev/shm/cga_" + session)
    my_ns = os.stat("/proc/self/ns/pid").st_ino
    abort, ns = 0, my_ns
    if creator in ("dead", "foreign"):
        p = subprocess.Popen([sys.executable, "-c", "pass"])
        p.wait()
        pid0, arrived = p.pid, world - 1
        if creator == "foreign":  # round 6 (advisor): the creator lives in ANOTHER pid namespace (a container per GPU over one /dev/shm): its pid means
            ns, arrived = my_ns + 1, 0  # nothing here and must not be read as "gone"; the segment is left only when the name stops leading to it
    else:
        pid0, arrived = os.getpid(), 0
        abort = 1 if creator == "aborted" else 0  # a crashed job left its abort flag set (pid recycled): not "a peer aborted", a stale segment
    # comm_shm_t: ready, size, bar_count, bar_gen, abort, attached, pid0, pad, pidns0 (64 bits), pad[6]
    header = struct.pack("<8IQ6I", 0x43474331, world, arrived, 0, abort, world - 1, pid0, 0, ns, *([0] * 6))
    path.write_bytes(header + bytes(64 * 4096))
    code = ("import ctypes as C, sys; from cugraph_amd import _capi as capi; l = capi.lib(); e = C.c_void_p(); "
            "rc = l.cugraph_amd_comm_host_selftest(sys.argv[1].encode(), int(sys.argv[2]), int(sys.argv[3]), 50, C.byref(e)); "
            "print(l.cugraph_error_message(e) if rc else 'ok'); sys.exit(rc)")
    env = dict(os.environ, CUGRAPH_AMD_COMM_TIMEOUT_S="30")
    start = lambda r: subprocess.Popen([sys.executable, "-c", code, session, str(r), str(world)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)  # noqa: E731
    try:
        procs = {r: start(r) for r in range(1, world)}
        time.sleep(1.5)  # the late rank 0
        procs[0] = start(0)
        outs = {r: p.communicate(timeout=120)[0] for r, p in procs.items()}
        for r, p in procs.items():
            assert p.returncode == 0, (r, outs[r][-2000:])
    finally:
        if path.exists():
            path.unlink()


def _assemble(tmp_path, world, nv):
    out = np.full(nv, np.nan, np.float64)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        assert np.isnan(out[z["v"]]).all(), "a vertex came back from two ranks"
        out[z["v"]] = z["x"]
    assert not np.isnan(out).any(), "every vertex must come back from exactly one rank"
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("world,weighted", [(1, "-"), (2, "-"), (3, "-"), (4, "w"), (8, "-")])
def test_mg_capi_pagerank(orc, tmp_path, world, weighted):
    """cugraph_graph_create_mg + cugraph_pagerank_allow_nonconvergence on a communicator handle, 1 .. 8 ranks sharing one GPU, against the
    single-process oracle at a fixed iteration count (1e-6 absolute, 2e-5 relative: the tolerance of the single-GPU parity tests)."""
    _pagerank_case(orc, tmp_path, world, weighted)


def _pagerank_case(orc, tmp_path, world, weighted, env=None):
    from test_mg import truth

    scale, iters = 12, 12
    res = run_ranks("pagerank", world, tmp_path, scale, iters, 0.0, weighted, env_extra=env)
    assert sum(r["rows"] for r in res) == 1 << scale and all(r["repeat_equal"] for r in res)
    pr = _assemble(tmp_path, world, 1 << scale)
    t, _, _ = truth(orc, scale, 0.0, iters, weighted=weighted == "w")
    assert np.max(np.abs(pr - t)) <= 1e-6
    assert np.max(np.abs(pr - t) / t) <= 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("world,weighted", [(1, "-"), (2, "w"), (3, "-"), (4, "-"), (6, "w"), (8, "-")])
def test_mg_capi_pagerank_2d_layout(orc, tmp_path, world, weighted):
    """Round 6: the reference's 2-D R x C layout (graph_view.hpp:64-230; 1x1, 1x2, 1x3, 2x2, 2x3, 2x4 for these worlds) behind the same two entry
    points, chosen with CUGRAPH_AMD_MG_LAYOUT=2d: x all-gathered over the column group, partial rows reduced over the row group to their owners, all
    as peer writes on the library's communicator.  Same oracle, same tolerance, every vertex from exactly one rank, two calls bit-identical."""
    _pagerank_case(orc, tmp_path, world, weighted, env={"CUGRAPH_AMD_MG_LAYOUT": "2d"})


@pytest.mark.gpu
@pytest.mark.parametrize("world,weighted", [(2, "-"), (4, "w"), (6, "-")])
def test_mg_capi_pagerank_2d_layout_hypersparse_blocks(orc, tmp_path, world, weighted):
    """DCSR in the place the reference uses it (structure_utils.cuh:139-195): the local block of the 2-D layout stored as DCSC
    (the library does so from C >= 4 column groups; forced here on smaller grids) must give the bits the plain block gives, and both the oracle's
    values.  The 8-rank case of test_mg_capi_pagerank_2d_layout (2 x 4) runs the form by default."""
    from test_mg import truth

    scale, iters = 12, 12
    prs = []
    for dcsr in ("1", "0"):
        res = run_ranks("pagerank", world, tmp_path, scale, iters, 0.0, weighted, env_extra={"CUGRAPH_AMD_MG_LAYOUT": "2d", "CUGRAPH_AMD_MG_DCSR": dcsr})
        assert sum(r["rows"] for r in res) == 1 << scale and all(r["repeat_equal"] for r in res)
        prs.append(_assemble(tmp_path, world, 1 << scale))
    assert np.array_equal(prs[0], prs[1])
    t, _, _ = truth(orc, scale, 0.0, iters, weighted=weighted == "w")
    assert np.max(np.abs(prs[0] - t)) <= 1e-6 and np.max(np.abs(prs[0] - t) / t) <= 2e-5


@pytest.mark.gpu
def test_mg_capi_pagerank_2d_layout_converges_like_single_gpu(orc, tmp_path):
    from test_mg import truth

    res = run_ranks("pagerank", 4, tmp_path, 11, 200, 1e-5, "-", env_extra={"CUGRAPH_AMD_MG_LAYOUT": "2d"})
    assert all(r["converged"] for r in res)
    pr = _assemble(tmp_path, 4, 1 << 11)
    t, it, tconv = truth(orc, 11, 1e-5, 200)
    assert tconv
    np.testing.assert_allclose(pr, t, rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("what", ["pagerank", "pagerank2d", "bfs", "sssp"])
def test_mg_capi_on_the_windows_of_a_multi_gpu_node(orc, tmp_path, what):
    """The exchange paths as they run when the ranks sit on DIFFERENT GPUs: the communicator then allocates its windows fine-grained (system-scope
    coherent), which ranks sharing one GPU never do by themselves.  CUGRAPH_AMD_COMM_WINDOWS=finegrained forces that memory kind: 3 ranks, every algorithm
    family, both PageRank layouts, the same oracles."""
    env = {"CUGRAPH_AMD_COMM_WINDOWS": "finegrained"}
    if what == "pagerank":
        _pagerank_case(orc, tmp_path, 3, "w", env=env)
    elif what == "pagerank2d":
        _pagerank_case(orc, tmp_path, 4, "-", env=dict(env, CUGRAPH_AMD_MG_LAYOUT="2d"))
    elif what == "bfs":
        _bfs_case(orc, tmp_path, 3, "", env=env)
    else:
        _sssp_case(orc, tmp_path, 3, "int", env=env)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_mg_capi_bad_argument_on_one_rank_fails_everywhere(tmp_path, world):
    """Round 6 (review of round 5, advisor): the rank-local argument checks of the collective entry points exchange their verdicts -- and the scalar
    arguments are compared -- BEFORE the first collective of the algorithm: alpha out of range on rank 0, a negative epsilon on the last rank, different
    iteration counts, different SSSP sources, different BFS depth limits: every rank gets an error at once (a lone rank used to leave its peers in a
    barrier for the communicator's 60 s timeout, which then poisoned the session), and the next valid call on the same communicator works."""
    res = run_ranks("agree", world, tmp_path, 10, timeout=120)
    for r in res:
        assert r["seconds"] < 20.0, r
        assert all(m != "accepted" for m in r["messages"].values()), r["messages"]
        assert "alpha" in r["messages"]["alpha"] or "rejected its arguments" in r["messages"]["alpha"]
        assert "different scalar arguments" in r["messages"]["iterations"] and "different scalar arguments" in r["messages"]["source"]
        for name in ("degrees", "paths"):  # (a wrongly typed vertex column on one rank)
            assert "must match" in r["messages"][name] or "rejected its arguments" in r["messages"][name], r["messages"]
    assert sum(r["rows_after"] for r in res) == 1 << 10


@pytest.mark.gpu
def test_mg_capi_pagerank_many_calls_reuse_channels(tmp_path):
    """Round 5 (advisor finding): 80 cugraph_pagerank calls on one communicator -- more than its 64 signal channels; the plans return theirs."""
    res = run_ranks("pagerank", 2, tmp_path, 10, 6, 0.0, "-", 80)
    assert all(r["repeat_equal"] and r["many_calls_equal"] for r in res)


@pytest.mark.gpu
def test_mg_capi_pagerank_converges_like_single_gpu(orc, tmp_path):
    from test_mg import truth

    res = run_ranks("pagerank", 2, tmp_path, 11, 200, 1e-5, "-")
    assert all(r["converged"] for r in res)
    pr = _assemble(tmp_path, 2, 1 << 11)
    t, it, tconv = truth(orc, 11, 1e-5, 200)
    assert tconv
    np.testing.assert_allclose(pr, t, rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 3])
def test_mg_capi_personalized_pagerank_with_guess_and_out_weights(orc, tmp_path, world):
    """cugraph_personalized_pagerank on a multi-GPU graph with all three optional (vertices, values) arguments, each rank passing a slice that
    names other ranks' vertices: against the oracle (personalization with a repeated vertex, un-normalised initial guess, precomputed
    out-weight sums) at a fixed iteration count; an id that is no vertex is refused on every rank."""
    _ppr_case(orc, tmp_path, world)


def _ppr_case(orc, tmp_path, world, env=None):
    from mg_capi_cases import ppr_inputs
    from test_mg import rmat_graph

    scale, iters = 11, 10
    res = run_ranks("ppr", world, tmp_path, scale, iters, env_extra=env)
    nv = 1 << scale
    assert sum(r["rows"] for r in res) == nv
    assert all("not in the graph" in r["bad_vertex"] for r in res), [r["bad_vertex"] for r in res]
    pr = _assemble(tmp_path, world, nv)
    s, d = rmat_graph(orc, scale)
    off, idx, ww = orc.coo_to_cs(nv, d, s, None)
    pv, pval, guess, outw = ppr_inputs(scale)
    t, _, _ = orc.pagerank(nv, off, idx, ww, 0.85, 0.0, iters, personalization=(pv, pval), initial_guess=guess, precomputed_outw=outw, acc64=True)
    assert np.max(np.abs(pr - t)) <= 1e-6
    nz = t > 1e-9
    assert np.max(np.abs(pr[nz] - t[nz]) / t[nz]) <= 5e-5


def _assemble_paths(tmp_path, world, nv, k):
    res = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    seen = np.zeros(nv, np.int32)
    dist = np.zeros(nv, res[0][f"d{k}"].dtype)
    pred = np.full(nv, -2, np.int32)
    for r in res:
        seen[r[f"v{k}"]] += 1
        dist[r[f"v{k}"]] = r[f"d{k}"]
        if f"p{k}" in r:
            pred[r[f"v{k}"]] = r[f"p{k}"]
    assert (seen == 1).all(), "every vertex must come back from exactly one rank"
    return res[0]["roots"], dist, pred


@pytest.mark.gpu
@pytest.mark.parametrize("world,direction", [(1, ""), (2, ""), (3, "topdown"), (4, "bottomup"), (4, "")])
def test_mg_capi_bfs(orc, tmp_path, world, direction):
    """cugraph_graph_create_mg + cugraph_bfs on a communicator handle (ranks sharing one GPU): distances bit-exact against the oracle,
    parents = minimum external id among the valid ones whatever directions the levels took; one rank names each source (the union of
    the ranks' lists is the source set), and one multi-source call with the whole list on every rank."""
    _bfs_case(orc, tmp_path, world, direction)


def _bfs_case(orc, tmp_path, world, direction, env=None):
    from test_mg_traversal import check_bfs

    scale, n_roots = 12, 3
    res = run_ranks("bfs", world, tmp_path, scale, n_roots, "p", env_extra=dict(env or {}, CUGRAPH_AMD_MG_BFS=direction))
    for k in range(n_roots):
        roots, dist, pred = _assemble_paths(tmp_path, world, 1 << scale, k)
        check_bfs(orc, scale, [int(roots[k])], dist, pred)
    if direction == "bottomup":
        assert res[0]["bottom_up_levels"] == res[0]["levels"]
    if direction == "topdown":
        assert res[0]["bottom_up_levels"] == 0
    assert all(r["first_handle_again"] for r in res)  # the cached plan served a second handle of the communicator and then the first one again
    roots, dist, _ = _assemble_paths(tmp_path, world, 1 << scale, "m")
    nv, s, d, _, off, idx, _ = __import__("test_mg_traversal").graph(orc, scale)
    od, _ = orc.bfs(nv, off, idx, np.asarray(roots, np.int32), 2**31 - 1)
    assert np.array_equal(dist, od)


@pytest.mark.gpu
@pytest.mark.parametrize("world,direction", [(2, "topdown"), (4, "")])
def test_mg_capi_bfs_count_matrix_through_the_host(orc, tmp_path, world, direction):
    """The default top-down level exchanges counts and tuples from the device (no host in it: k_push_tuples_dev, the reference's device all-to-all of
    shuffle_comm.cuh:139-186); CUGRAPH_AMD_MG_BFS_DEVICE_EXCHANGE=0 is round 5's exchange (count matrix all-gathered by the hosts): same answers."""
    _bfs_case(orc, tmp_path, world, direction, env={"CUGRAPH_AMD_MG_BFS_DEVICE_EXCHANGE": "0"})


@pytest.mark.gpu
def test_mg_capi_bfs_depth_limit(orc, tmp_path):
    from test_mg_traversal import check_bfs

    scale = 12
    run_ranks("bfs", 2, tmp_path, scale, 1, "p", 2)
    roots, dist, pred = _assemble_paths(tmp_path, 2, 1 << scale, 0)
    check_bfs(orc, scale, [int(roots[0])], dist, pred, depth_limit=2)


@pytest.mark.gpu
@pytest.mark.parametrize("world,kind", [(1, "int"), (2, "int"), (3, "int"), (4, "unit")])
def test_mg_capi_sssp(orc, tmp_path, world, kind):
    """cugraph_sssp on a multi-GPU graph: distances bit-identical to Dijkstra (integer and unit weights), minimum-external-id parents"""
    _sssp_case(orc, tmp_path, world, kind)


def _sssp_case(orc, tmp_path, world, kind, env=None):
    from test_mg_traversal import check_sssp, graph

    scale, n_roots = 12, 2
    run_ranks("sssp", world, tmp_path, scale, n_roots, "p", kind, env_extra=env)
    for k in range(n_roots):
        roots, dist, pred = _assemble_paths(tmp_path, world, 1 << scale, k)
        if kind == "int":
            check_sssp(orc, scale, int(roots[k]), dist, pred)
        else:
            nv, s, d, _, off, idx, _ = graph(orc, scale)
            od, _ = orc.bfs(nv, off, idx, np.asarray([roots[k]], np.int32), 2**31 - 1)
            want = np.where(od == 2**31 - 1, np.finfo(np.float32).max, od.astype(np.float32)).astype(np.float32)
            assert np.array_equal(dist.view(np.uint32), want.view(np.uint32))  # unit weights: the BFS distances, bit for bit


@pytest.mark.gpu
@pytest.mark.parametrize("world,env", [(2, {"CUGRAPH_AMD_MG_SSSP_WINDOW": "1"}), (3, {"CUGRAPH_AMD_MG_SSSP_WINDOW": "1", "CUGRAPH_AMD_SSSP_DELTA_SCALE": "0.1"}),
                                       (2, {"CUGRAPH_AMD_MG_SSSP_WINDOW": "1", "CUGRAPH_AMD_MG_SSSP_INPLACE": "0"})])
def test_mg_capi_sssp_schedules(orc, tmp_path, world, env):
    """the opt-in near / far windows of the partitioned SSSP do not change the result: windows of the reference's width, windows a tenth as
    wide (many window moves, far piles split again and again), every candidate through the exchange (owner-side near / far decision only) --
    distances bit-identical to Dijkstra, minimum-external-id parents, with a cut-off (the default -- one unbounded window -- runs under every
    other SSSP test of this file)"""
    from test_mg_traversal import check_sssp

    scale, n_roots = 12, 2
    run_ranks("sssp", world, tmp_path, scale, n_roots, "p", "int", 900.0, env_extra=env)
    for k in range(n_roots):
        roots, dist, pred = _assemble_paths(tmp_path, world, 1 << scale, k)
        check_sssp(orc, scale, int(roots[k]), dist, pred, cutoff=900.0)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 3])
def test_mg_capi_sssp_float64(orc, tmp_path, world):
    """cugraph_sssp on a multi-GPU graph with FLOAT64 weights (integer + 1/3: not float32 values): distances equal Dijkstra's in double to the last
    bit, DBL_MAX where unreached, minimum-external-id parents among the tight in-edges"""
    from test_mg_traversal import graph, min_ext_parent

    scale, n_roots = 11, 2
    run_ranks("sssp", world, tmp_path, scale, n_roots, "p", "f64")
    nv, s, d, w, off, idx, ww = graph(orc, scale, weighted=True)
    w64, ww64 = w.astype(np.float64) + 1.0 / 3.0, ww.astype(np.float64) + 1.0 / 3.0
    for k in range(n_roots):
        roots, dist, pred = _assemble_paths(tmp_path, world, nv, k)
        assert dist.dtype == np.float64
        od = dijkstra_f64(nv, off, idx, ww64, int(roots[k]))
        assert np.array_equal(dist, od)
        dmax = np.finfo(np.float64).max
        ok = (od[s] != dmax) & (od[d] != dmax) & (od[s] + w64 == od[d])
        want = min_ext_parent(nv, s, d, ok)
        want[int(roots[k])] = -1
        assert np.array_equal(pred, want)


def dijkstra_f64(nv, off, idx, w, source):
    """plain binary-heap Dijkstra in double (checker for the FLOAT64 case; the distances of a shortest-path tree do not depend on the order
    relaxations are tried in: min over paths of left-to-right sums along the path -- every algorithm that only ever forms dist[u] + w(u, v) agrees)"""
    import heapq

    dist = np.full(nv, np.finfo(np.float64).max)
    dist[source] = 0.0
    heap = [(0.0, source)]
    while heap:
        du, u = heapq.heappop(heap)
        if du > dist[u]:
            continue
        for p in range(off[u], off[u + 1]):
            nd = du + w[p]
            v = idx[p]
            if nd < dist[v]:
                dist[v] = nd
                heapq.heappush(heap, (nd, v))
    return dist


def _assemble_clusters(tmp_path, world, nv):
    c = np.full(nv, -1, np.int64)
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        assert (c[z["v"]] == -1).all(), "a vertex came back from two ranks"
        c[z["v"]] = z["c"]
    assert (c >= 0).all(), "every vertex must come back from exactly one rank"
    return c.astype(np.int32)


@pytest.mark.gpu
@pytest.mark.parametrize("world,scale", [(1, 14), (2, 14), (3, 14), (4, 16)])
def test_mg_capi_louvain(orc, tmp_path, world, scale):
    """cugraph_louvain on a multi-GPU graph (BASELINE config 5's algorithm partitioned over ranks sharing one GPU): the clustering equals the C
    oracle's -- and therefore the single-GPU library's, which the parity suite pins to the same oracle -- vertex for vertex, the modularity is
    the same double on every rank and within 1e-9 of the oracle's."""
    _louvain_case(orc, tmp_path, world, scale)


def _louvain_case(orc, tmp_path, world, scale, env=None):
    from test_gpu_parity import louvain_rmat_input

    res = run_ranks("louvain", world, tmp_path, scale, env_extra=env)
    nv = 1 << scale
    assert sum(r["rows"] for r in res) == nv and len({r["modularity_hex"] for r in res}) == 1
    c = _assemble_clusters(tmp_path, world, nv)
    src, dst, w = louvain_rmat_input(orc, scale)
    oc, oq, _, osweeps = orc.louvain_c(nv, src, dst, w, 100, 1e-7, 1.0)
    q = float.fromhex(res[0]["modularity_hex"])
    assert abs(q - oq) <= 1e-9
    assert np.array_equal(c, oc)
    assert res[0]["sweeps"] == osweeps


@pytest.mark.gpu
def test_mg_capi_louvain_rmat22_golden(tmp_path):
    """the partitioned run at the size the single-GPU timing is quoted on, two ranks, against the committed fixture of the C oracle
    (tests/golden/louvain_rmat22.json): cluster column (sha256), modularity to 1e-9 (and the same double on both ranks)"""
    import hashlib

    gold = json.loads((ROOT / "tests" / "golden" / "louvain_rmat22.json").read_text())
    res = run_ranks("louvain", 2, tmp_path, gold["scale"], timeout=900)
    c = _assemble_clusters(tmp_path, 2, 1 << gold["scale"])
    assert abs(float.fromhex(res[0]["modularity_hex"]) - gold["modularity"]) <= 1e-9 and res[0]["modularity_hex"] == res[1]["modularity_hex"]
    assert int(np.unique(c).size) == gold["clusters"]
    assert hashlib.sha256(np.ascontiguousarray(c, np.int32).tobytes()).hexdigest() == gold["clusters_sha256"]


# ----------------------------------------------------------------------- round 5: the multi-GPU graph at the width of the single-GPU one
WIDE = {"CUGRAPH_AMD_TEST_WIDE_IDS": "1"}


@pytest.mark.gpu
@pytest.mark.parametrize("what,world", [("pagerank", 2), ("ppr", 3), ("bfs", 2), ("sssp", 3), ("sssp", 1), ("louvain", 2)])
def test_mg_capi_int64_vertex_ids(orc, tmp_path, what, world):
    """INT64 vertex ids on a graph from cugraph_graph_create_mg (graph_mg.cpp:127-151 instantiates vertex_t = int64_t; python-cugraph's default):
    every id crosses the C API as v * 1000003 + 2^40.  The ranks agree on one sorted id list (mg_graph.hip: mg_outer_ids), the engines run on
    compact int32 ids, and every vertex column that comes back -- result vertices, predecessors, BFS hop counts typed like the vertices -- is in
    the caller's id space: mapped home by the worker, the results pass the very checks of the int32 cases (oracle parity, minimum-external-id
    parents: the mapping is monotone)."""
    if what == "pagerank":
        _pagerank_case(orc, tmp_path, world, "-", env=WIDE)
    elif what == "ppr":
        _ppr_case(orc, tmp_path, world, env=WIDE)
    elif what == "bfs":
        _bfs_case(orc, tmp_path, world, "", env=WIDE)
    elif what == "sssp":
        _sssp_case(orc, tmp_path, world, "int", env=WIDE)
    else:
        _louvain_case(orc, tmp_path, world, 14, env=WIDE)


@pytest.mark.gpu
@pytest.mark.parametrize("world,id_kind,wide", [(1, "i32", False), (2, "i64", False), (3, "i32", True)])
def test_mg_capi_edge_properties_and_decompress(orc, tmp_path, world, id_kind, wide):
    """Edge ids / edge type ids handed to cugraph_graph_create_mg stay with their edges and come back from cugraph_decompress_to_edgelist, every
    rank returning its part: the union over the ranks is the input edge list, and the id next to an edge names exactly that input edge
    (graph_mg.cpp:127-151, decompress_to_edgelist.cpp:60-103).  On the same graph: cugraph_degrees (all vertices / a listed few) against bincounts,
    cugraph_has_vertex; with drop_self_loops the properties are refused on every rank (as cugraph_graph_create_sg refuses them)."""
    from test_mg import rmat_graph

    scale = 10
    res = run_ranks("props", world, tmp_path, scale, id_kind, env_extra=WIDE if wide else None)
    nv, ne = 1 << scale, 16 << scale
    s, d = rmat_graph(orc, scale)
    w = np.random.default_rng(1).integers(1, 9, size=ne).astype(np.float32)
    ids = (np.random.default_rng(4).permutation(ne) + 1000).astype(np.int64 if id_kind == "i64" else np.int32)
    types = (np.arange(ne) % 5).astype(np.int32)
    z = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    es, ed, ew, ei, et = (np.concatenate([x[k] for x in z]) for k in ("s", "d", "w", "ids", "types"))
    assert ei.dtype == ids.dtype and es.size == ne and sorted(ei.tolist()) == sorted(ids.tolist())
    where = np.empty(ne + 1000, np.int64)
    where[ids] = np.arange(ne)
    k = where[ei]  # the input edge every returned row claims to be
    assert np.array_equal(es, s[k]) and np.array_equal(ed, d[k]) and np.array_equal(ew, w[k]) and np.array_equal(et, types[k])
    din, dout = np.bincount(d, minlength=nv), np.bincount(s, minlength=nv)
    v = np.concatenate([x["v"] for x in z])
    assert np.array_equal(np.sort(v), np.arange(nv))
    assert np.array_equal(np.concatenate([x["din"] for x in z]), din[v]) and np.array_equal(np.concatenate([x["dout"] for x in z]), dout[v])
    v2 = np.concatenate([x["v2"] for x in z])
    assert np.array_equal(np.sort(v2), np.sort(np.concatenate([x["listed"] for x in z])))
    assert np.array_equal(np.concatenate([x["din2"] for x in z]), din[v2]) and np.array_equal(np.concatenate([x["dout2"] for x in z]), dout[v2])
    assert all(r["has_vertex"] == [True, True, False] for r in res)
    assert all("not supported" in r["refused"] or "NOT_IMPLEMENTED" in r["refused"] for r in res), [r["refused"] for r in res]


@pytest.mark.gpu
@pytest.mark.parametrize("world,wide", [(1, False), (2, False), (3, True)])
def test_mg_capi_extract_paths(orc, tmp_path, world, wide):
    """cugraph_bfs (with predecessors) + cugraph_extract_paths on a multi-GPU graph (extract_paths.cpp:57-119, extract_bfs_paths_impl.cuh:130-240
    with multi_gpu = true): every rank asks for its own destinations, most of them owned by other ranks.  Row i is the root ... destination i along
    edges of the graph in exactly dist + 1 entries, -1 beyond; a destination the BFS did not reach gives a row of -1; the matrix is as wide on
    every rank (the longest path over all ranks' destinations + 1); a destination that is no vertex is INVALID_INPUT on every rank."""
    from test_mg_traversal import graph

    scale = 11
    res = run_ranks("paths", world, tmp_path, scale, env_extra=WIDE if wide else None)
    nv, s, d, _, off, idx, _ = graph(orc, scale)
    z = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    root = int(z[0]["root"])
    od, _ = orc.bfs(nv, off, idx, np.asarray([root], np.int32), 2**31 - 1)
    edges = set(zip(s.tolist(), d.tolist()))
    unreached = 2**31 - 1
    longest = max((int(od[x["dests"]][od[x["dests"]] != unreached].max()) if (od[x["dests"]] != unreached).any() else 0) for x in z)
    assert all(r["width"] == longest + 1 for r in res), ([r["width"] for r in res], longest)
    n_unreached = 0
    for x in z:
        assert x["paths"].shape == (x["dests"].size, longest + 1)
        for dest, row in zip(x["dests"].tolist(), x["paths"].tolist()):
            if od[dest] == unreached:
                assert all(p == -1 for p in row)
                n_unreached += 1
                continue
            n = int(od[dest]) + 1
            assert row[0] == root and row[n - 1] == dest and all(p == -1 for p in row[n:])
            assert all((a, b) in edges for a, b in zip(row[:n - 1], row[1:n]))
    assert n_unreached > 0  # (RMAT: a third of the ids have no in-edge)
    assert all("not in the graph" in r["bad_destination"] for r in res), [r["bad_destination"] for r in res]
