"""GPU parity tests: the HIP path, called through the C ABI (cugraph_amd.pylib -> libcugraph_c.so), against
  * the reference's golden vectors (tests/golden/golden.json),
  * the CPU oracle (oracle/) on the same seeded RMAT edge lists,
  * size-independent properties at larger sizes.
Tolerances: BFS / SSSP distances and parents bit-exact; PageRank |a-b| <= 1e-6 absolute (north_star) AND
<= 2e-5 relative to the fp64-accumulating oracle at a fixed iteration count (the reference's own tolerance is
1e-3 relative, cpp/tests/link_analysis/pagerank_test.cpp:328-334)."""
import numpy as np
import pytest

from conftest import int_weights, rmat_graph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cg():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import cugraph_amd

    return cugraph_amd


@pytest.fixture(scope="module")
def handle(cg):
    return cg.ResourceHandle()


def T(a, dtype=None):
    import torch

    return torch.as_tensor(np.ascontiguousarray(a, dtype=dtype), device="cuda")


def make_graph(cg, handle, src, dst, wgt=None, transposed=False, renumber=False, symmetric=False, vertices=None, wdtype=np.float32):
    props = cg.GraphProperties(is_symmetric=symmetric, is_multigraph=True)
    return cg.SGGraph(handle, props, T(src, np.int32), T(dst, np.int32), None if wgt is None else T(wgt, wdtype),
                      store_transposed=transposed, renumber=renumber, vertices_array=None if vertices is None else T(vertices, np.int32))


def by_vertex(verts, *cols):
    """results come in internal order with the external id column: scatter back to external order"""
    v = verts.cpu().numpy()
    out = []
    for c in cols:
        c = c.cpu().numpy()
        if c.size == 0:
            out.append(c)
            continue
        r = np.empty_like(c)
        r[v] = c
        out.append(r)
    return out


def bfs_expected_parents(src, dst, dist_ext, verts):
    """The library's parent rule: among the in-neighbours one level up, the one with the smallest INTERNAL id
    (internal id of external vertex x = position of x in the result's vertex column)."""
    nv = dist_ext.size
    verts = verts.cpu().numpy().astype(np.int64)
    int_of = np.empty(nv, np.int64)
    int_of[verts] = np.arange(nv)
    ds, dd = dist_ext[src].astype(np.int64), dist_ext[dst].astype(np.int64)
    ok = (ds != np.iinfo(np.int32).max) & (dd == ds + 1)
    best = np.full(nv, np.iinfo(np.int64).max)
    np.minimum.at(best, dst[ok], int_of[src[ok]])
    pred = np.full(nv, -1, np.int32)
    has = best != np.iinfo(np.int64).max
    pred[has] = verts[best[has]]
    return pred


def nearly_equal(a, b, eps):
    return abs(a - b) <= max(abs(a), abs(b)) * eps


# ------------------------------------------------------------------------------ reference goldens
def test_capi_pagerank_goldens(cg, handle, golden):
    for case in golden["c_api"]["pagerank"]:
        gr = case["graph"]
        for renumber in (False, True):
            g = make_graph(cg, handle, gr["src"], gr["dst"], gr["wgt"], transposed=case["store_transposed"], renumber=renumber)
            v, pr, conv = cg.pagerank(handle, g, None, None, None, None, case["alpha"], case["epsilon"], case["max_iterations"], False,
                                      fail_on_nonconvergence=False)
            assert conv == case["converged"], case["name"]
            (pr,) = by_vertex(v, pr)
            for a, b in zip(pr, case["result"]):
                assert nearly_equal(float(a), b, golden["c_api"]["tolerance"]), (case["name"], pr)
            if not case["converged"]:
                with pytest.raises(cg.FailedToConvergeError):
                    cg.pagerank(handle, g, None, None, None, None, case["alpha"], case["epsilon"], case["max_iterations"], False)


def test_capi_personalized_goldens(cg, handle, golden):
    for case in golden["c_api"]["personalized_pagerank"]:
        gr = case["graph"]
        g = make_graph(cg, handle, gr["src"], gr["dst"], gr["wgt"], transposed=case["store_transposed"])
        v, pr, conv = cg.personalized_pagerank(handle, g, None, None, None, None, T(case["pers_vertices"], np.int32),
                                               T(case["pers_values"], np.float32), case["alpha"], case["epsilon"],
                                               case["max_iterations"], False, fail_on_nonconvergence=False)
        assert conv == case["converged"]
        (pr,) = by_vertex(v, pr)
        for a, b in zip(pr, case["result"]):
            assert nearly_equal(float(a), b, golden["c_api"]["tolerance"]), (case["name"], pr)


def test_pylibcugraph_pagerank_goldens(cg, handle, golden):
    p = golden["pylibcugraph_pagerank"]["params"]
    for name in ("karate.csv", "dolphins.csv", "Simple_1", "Simple_2"):
        gr = golden["graphs"][name]
        g = make_graph(cg, handle, gr["src"], gr["dst"], gr["wgt"], transposed=True)  # conftest.py: renumber=False
        v, pr = cg.pagerank(handle, g, None, None, None, None, p["alpha"], p["epsilon"], p["max_iterations"], False)
        exp = golden["pylibcugraph_pagerank"][name]
        assert v.cpu().numpy().tolist() == exp["vertex"]  # renumber=False: identity numbering
        np.testing.assert_allclose(pr.cpu().numpy(), np.array(exp["pagerank"]), rtol=p["rel_tol"], atol=5e-7)


def test_capi_bfs_goldens(cg, handle, golden):
    for case in golden["c_api"]["bfs"]:
        gr = case["graph"]
        for renumber in (False, True):
            g = make_graph(cg, handle, gr["src"], gr["dst"], gr["wgt"], transposed=case["store_transposed"], renumber=renumber)
            d, p, v = cg.bfs(handle, g, T(case["seeds"], np.int32), False, case["depth_limit"], True, False)
            d, p = by_vertex(v, d, p)
            assert d.tolist() == case["distances"]
            assert p.tolist() == case["predecessors"]


def test_bfs_exceptions(cg, handle, golden):
    """bfs_test.c:108-158 test_bfs_exceptions: INT64 seeds on an INT32 graph -> CUGRAPH_INVALID_INPUT."""
    import ctypes as C

    from cugraph_amd import _capi

    gr = golden["c_api"]["bfs"][0]["graph"]
    g = make_graph(cg, handle, gr["src"], gr["dst"], gr["wgt"])
    seeds = T([0], np.int64)
    l = _capi.lib()
    view = l.cugraph_type_erased_device_array_view_create(C.c_void_p(seeds.data_ptr()), 1, _capi.INT64)
    res, err = C.c_void_p(), C.c_void_p()
    code = l.cugraph_bfs(handle.c_resource_handle_ptr, g.c_graph_ptr, view, 0, 1, 1, 0, C.byref(res), C.byref(err))
    assert code == _capi.CUGRAPH_INVALID_INPUT
    assert b"vertex type of graph and sources must match" in l.cugraph_error_message(err)
    l.cugraph_error_free(err)
    l.cugraph_type_erased_device_array_view_free(view)
    with pytest.raises(ValueError):  # bfs.pyx:140-143: source that is not a vertex
        cg.bfs(handle, g, T([17], np.int32), False, 0, True, False)


def test_capi_sssp_goldens(cg, handle, golden):
    for case in golden["c_api"]["sssp"]:
        gr = case["graph"]
        dt = np.dtype(case["dtype"])
        for renumber in (False, True):
            g = make_graph(cg, handle, gr["src"], gr["dst"], gr["wgt"], transposed=case["store_transposed"], renumber=renumber, wdtype=dt)
            v, d, p = cg.sssp(handle, g, case["source"], float(np.finfo(dt).max), True, False)
            d, p = by_vertex(v, d, p)
            assert d.dtype == dt
            for a, b in zip(d, case["distances"]):
                assert nearly_equal(float(a), b, golden["c_api"]["tolerance"])
            assert p.tolist() == case["predecessors"]


def test_pylibcugraph_sssp_goldens(cg, handle, golden):
    for name, exp in golden["pylibcugraph_sssp"].items():
        gr = golden["graphs"][name]
        g = make_graph(cg, handle, gr["src"], gr["dst"], gr["wgt"])
        v, d, p = cg.sssp(handle, g, exp["start_vertex"], float(np.finfo(np.float32).max), True, False)
        assert v.cpu().numpy().tolist() == exp["vertex"]
        np.testing.assert_allclose(d.cpu().numpy(), np.array(exp["distance"], np.float32), rtol=1e-4)
        if exp["check_predecessor"]:
            assert p.cpu().numpy().tolist() == exp["predecessor"]


def test_karate_vs_networkx(cg, handle, golden):
    """BASELINE.json configs[0]: karate.csv PageRank + BFS against NetworkX."""
    nx = pytest.importorskip("networkx")
    gr = golden["graphs"]["karate.csv"]
    G = nx.DiGraph()
    G.add_weighted_edges_from(zip(gr["src"], gr["dst"], gr["wgt"]))
    g = make_graph(cg, handle, gr["src"], gr["dst"], gr["wgt"], transposed=True, renumber=True)
    v, pr = cg.pagerank(handle, g, None, None, None, None, 0.85, 1e-7, 500, False)
    (pr,) = by_vertex(v, pr)
    ref = nx.pagerank(G, alpha=0.85, tol=1e-10, max_iter=1000)
    np.testing.assert_allclose(pr, np.array([ref[i] for i in range(34)]), rtol=1e-4)
    d, p, v = cg.bfs(handle, g, T([0], np.int32), False, 0, True, False)
    (d,) = by_vertex(v, d)
    sp = nx.single_source_shortest_path_length(G, 0)
    assert d.tolist() == [sp[i] for i in range(34)]


# ------------------------------------------------------------------------------ invalid inputs
def test_graph_creation_errors(cg, handle):
    props = cg.GraphProperties()
    with pytest.raises(ValueError, match="src size != dst size"):  # conftest InvalidNumVerts_1
        cg.SGGraph(handle, props, T([1, 2], np.int32), T([1, 2, 3], np.int32), T([1, 1, 1], np.float32))
    with pytest.raises(ValueError, match="src size != weights size"):  # conftest InvalidNumWeights_1
        cg.SGGraph(handle, props, T([0, 1, 2], np.int32), T([1, 2, 3], np.int32), T([1, 1, 1, 1], np.float32))
    with pytest.raises(TypeError):
        cg.SGGraph(handle, props, [0, 1], T([1, 2], np.int32))
    g64 = cg.SGGraph(handle, props, T([0, 1], np.int64), T([1, 2], np.int64))  # INT64 ids (graph_sg.cpp:745-779): accepted
    with pytest.raises(ValueError, match="vertex type of graph and (sources|vertices) must match"):  # (bfs.pyx:140 checks has_vertex first)
        cg.bfs(handle, g64, T([0], np.int32), False, 0, True, False)
    # edge ids / types are validated edge properties that no algorithm of this library reads (graph.hip): accepted, sizes checked
    cg.SGGraph(handle, props, T([0, 1], np.int32), T([1, 2], np.int32), edge_id_array=T([0, 1], np.int32))
    with pytest.raises(ValueError, match="edge id prop size"):
        cg.SGGraph(handle, props, T([0, 1], np.int32), T([1, 2], np.int32), edge_id_array=T([0, 1, 2], np.int32))
    g = cg.SGGraph(handle, props, T([0, 1], np.int32), T([1, 2], np.int32))  # unweighted
    with pytest.raises(ValueError, match="weighted"):
        cg.sssp(handle, g, 0, 1e30, True, False)
    hv = cg.has_vertex(handle, g, T([0, 2, 3, -1], np.int32))
    assert hv.cpu().numpy().tolist() == [True, True, False, False]


def test_empty_and_isolated(cg, handle):
    props = cg.GraphProperties()
    # vertices 0..9 given explicitly, edges only among 0..2: isolated vertices keep base rank
    g = cg.SGGraph(handle, props, T([0, 1], np.int32), T([1, 2], np.int32), T([1, 1], np.float32), store_transposed=True,
                   vertices_array=T(np.arange(10), np.int32))
    assert g.num_vertices == 10 and g.num_edges == 2
    v, pr, conv = cg.pagerank(handle, g, None, None, None, None, 0.85, 1e-8, 200, False, fail_on_nonconvergence=False)
    pr = pr.cpu().numpy()
    assert abs(pr.sum() - 1.0) < 1e-5
    d, p, v = cg.bfs(handle, g, T([0], np.int32), False, 0, True, False)
    assert d.cpu().numpy().tolist() == [0, 1, 2] + [2147483647] * 7
    # an edgeless graph
    g0 = cg.SGGraph(handle, props, T([], np.int32), T([], np.int32), T([], np.float32), store_transposed=True,
                    vertices_array=T(np.arange(4), np.int32))
    v, pr, conv = cg.pagerank(handle, g0, None, None, None, None, 0.85, 1e-8, 10, False, fail_on_nonconvergence=False)
    np.testing.assert_allclose(pr.cpu().numpy(), 0.25, rtol=1e-6)


# ------------------------------------------------------------------------------ RMAT vs the oracle
@pytest.mark.parametrize("scale,weighted,renumber,transposed", [
    (12, False, True, True), (12, True, True, True), (14, False, False, True), (16, False, True, True),
    (16, True, True, False), (18, False, True, True)])
def test_pagerank_rmat_vs_oracle(cg, handle, orc, scale, weighted, renumber, transposed):
    s, d = rmat_graph(orc, scale)
    nv = 1 << scale
    w = int_weights(s.size) if weighted else None
    # the HIP generator must produce the same edge list
    gs, gd = cg.generate_rmat_edgelist(handle, scale, s.size)
    assert np.array_equal(gs.cpu().numpy(), s) and np.array_equal(gd.cpu().numpy(), d)
    g = make_graph(cg, handle, s, d, w, transposed=transposed, renumber=renumber, vertices=np.arange(nv))
    iters = 20
    v, pr, conv = cg.pagerank(handle, g, None, None, None, None, 0.85, 0.0, iters, False, fail_on_nonconvergence=False)
    assert not conv
    (pr,) = by_vertex(v, pr)
    off, idx, ww = orc.coo_to_cs(nv, d, s, w)
    truth, it, _ = orc.pagerank(nv, off, idx, ww, 0.85, 0.0, iters, acc64=True)
    assert it == iters
    assert np.max(np.abs(pr - truth)) <= 1e-6
    rel = np.max(np.abs(pr - truth) / np.maximum(truth, 1e-30))
    assert rel <= 2e-5, rel
    assert abs(float(pr.astype(np.float64).sum()) - 1.0) < 1e-4  # mass conservation (dangling mass redistributed)


@pytest.mark.parametrize("transposed,first_frac,weighted", [(True, 0.0, False), (True, 0.4, True), (False, 0.0, True), (False, 1.0, False), (True, 0.07, False)])
def test_hypersparse_rows_vs_oracle(cg, handle, orc, transposed, first_frac, weighted):
    """DCSR / DCSC (the reference's compress_hypersparse_offsets, structure_utils.cuh:139-195): a renumber = FALSE graph over a
    sparse id range is put into the CSR + DCSR hybrid form; (nzd rows, offsets) must equal the oracle's restatement bit for bit, the consumers that
    walk the form directly (PageRank's re-blocking, the degree calls, decompress_to_edgelist) must give what the plain graph gives -- PageRank bit for
    bit -- and an algorithm that needs plain offsets (BFS) must re-inflate it and agree as well."""
    rng = np.random.default_rng(5)
    nv, ne = 200_000, 40_000  # most rows are empty
    hubs = rng.integers(0, nv, 300)
    s = np.where(rng.random(ne) < 0.5, rng.choice(hubs, ne), rng.integers(0, nv, ne)).astype(np.int32)
    d = np.where(rng.random(ne) < 0.3, rng.choice(hubs, ne), rng.integers(0, nv, ne)).astype(np.int32)
    w = int_weights(ne) if weighted else None
    first = int(first_frac * nv)
    plain = make_graph(cg, handle, s, d, w, transposed=transposed, renumber=False, vertices=np.arange(nv))
    g = make_graph(cg, handle, s, d, w, transposed=transposed, renumber=False, vertices=np.arange(nv))
    g.compress_hypersparse(transposed, first)
    is_h, got_first, nzd, off = g.hypersparse_view(transposed)
    assert is_h and got_first == first
    major, minor = (d, s) if transposed else (s, d)
    p_off, _, _ = orc.coo_to_cs(nv, major, minor, w)
    want_off, want_nzd = orc.compress_hypersparse_offsets(p_off, first)
    assert np.array_equal(nzd.cpu().numpy(), want_nzd)
    assert np.array_equal(off.cpu().numpy().astype(np.int64), want_off)
    assert off.numel() < nv // 2 or first_frac >= 0.4  # the point of the form
    assert np.array_equal(orc.inflate_hypersparse_offsets(want_off, want_nzd, first, nv), np.asarray(p_off, np.int64))
    # consumers that walk the hybrid form
    for a, b in zip(cg.degrees(handle, g), cg.degrees(handle, plain)):
        assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())
    assert g.hypersparse_view(transposed)[0], "the degree calls must not re-inflate the orientation"
    e1, e2 = cg.decompress_to_edgelist(handle, g), cg.decompress_to_edgelist(handle, plain)
    key = lambda e: np.sort(e[0].cpu().numpy().astype(np.int64) * nv + e[1].cpu().numpy())  # noqa: E731
    assert np.array_equal(key(e1), key(e2)) and np.array_equal(key(e1), np.sort(s.astype(np.int64) * nv + d))
    v1, pr1, _ = cg.pagerank(handle, g, None, None, None, None, 0.85, 0.0, 15, False, fail_on_nonconvergence=False)
    v2, pr2, _ = cg.pagerank(handle, plain, None, None, None, None, 0.85, 0.0, 15, False, fail_on_nonconvergence=False)
    assert np.array_equal(v1.cpu().numpy(), v2.cpu().numpy()) and np.array_equal(pr1.cpu().numpy(), pr2.cpu().numpy())
    if transposed:
        assert g.hypersparse_view(True)[0], "PageRank's re-blocking must walk the hybrid form, not re-inflate it"
    (pr,) = by_vertex(v1, pr1)
    c_off, c_idx, c_w = orc.coo_to_cs(nv, d, s, w)
    truth, _, _ = orc.pagerank(nv, c_off, c_idx, c_w, 0.85, 0.0, 15, acc64=True)
    assert np.max(np.abs(pr - truth)) <= 1e-6 and np.max(np.abs(pr - truth) / truth) <= 2e-5
    # ... and one that reads plain offsets
    src0 = int(s[0])
    r1 = cg.bfs(handle, g, T([src0], np.int32), False, 0, True, False)
    r2 = cg.bfs(handle, plain, T([src0], np.int32), False, 0, True, False)
    for a, b in zip(r1, r2):
        assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())
    assert not g.hypersparse_view(False)[0]


def test_mggraph_one_rank_equals_sggraph(cg, handle, orc):
    """pylibcugraph's MGGraph (graphs.pyx:357-700) on a one-rank handle: the rank's slice as three arrays per column
    (cugraph_graph_create_with_times_mg concatenates them, always renumbers): PageRank, BFS and SSSP equal the oracle's."""
    scale = 12
    s, d = rmat_graph(orc, scale)
    nv = 1 << scale
    w = int_weights(s.size)
    cut = [0, s.size // 3, s.size // 2, s.size]
    parts = lambda a, t: [T(a[cut[i]: cut[i + 1]], t) for i in range(3)]  # noqa: E731
    props = cg.GraphProperties(is_multigraph=True)
    with pytest.raises(ValueError):
        cg.MGGraph(handle, props, parts(s, np.int32), parts(d, np.int32)[:2], num_arrays=3)
    g = cg.MGGraph(handle, props, parts(s, np.int32), parts(d, np.int32), parts(w, np.float32), store_transposed=False, num_arrays=3,
                   vertices_array=[T(np.arange(nv), np.int32), T(np.zeros(0), np.int32), T(np.zeros(0), np.int32)])
    assert g.num_vertices == nv and g.num_edges == s.size
    v, pr, _ = cg.pagerank(handle, g, None, None, None, None, 0.85, 0.0, 20, False, fail_on_nonconvergence=False)
    (pr,) = by_vertex(v, pr)
    off, idx, ww = orc.coo_to_cs(nv, d, s, w)
    truth, _, _ = orc.pagerank(nv, off, idx, ww, 0.85, 0.0, 20, acc64=True)
    assert np.max(np.abs(pr - truth)) <= 1e-6
    off, idx, ww = orc.coo_to_cs(nv, s, d, w)
    src = int(np.nonzero(np.diff(off) > 0)[0][1])
    v, dist, _ = cg.sssp(handle, g, src, float(np.finfo(np.float32).max), False, False)
    (dist,) = by_vertex(v, dist)
    assert np.array_equal(dist, orc.sssp(nv, off, idx, ww, src)[0])
    dist, _, v = cg.bfs(handle, g, T(np.array([src]), np.int32), False, 0, False, False)
    (dist,) = by_vertex(v, dist)
    assert np.array_equal(dist, orc.bfs(nv, off, idx, np.array([src], np.int32))[0])


def test_pagerank_converged_iterations_and_initial_guess(cg, handle, orc):
    scale = 13
    s, d = rmat_graph(orc, scale, seed=5)
    nv = 1 << scale
    g = make_graph(cg, handle, s, d, None, transposed=True, renumber=True, vertices=np.arange(nv))
    off, idx, _ = orc.coo_to_cs(nv, d, s)
    truth, it, conv = orc.pagerank(nv, off, idx, None, 0.85, 1e-5, 500, acc64=True)
    plan = cg.PageRankPlan(handle, g, 0.85)
    done, c = plan.step(500, epsilon=1e-5)
    assert c and abs(done - it) <= 1  # fp32 vs fp64 L1 sums may flip the last comparison
    # warm start from the converged vector: one iteration suffices, result unchanged (idempotence)
    v, pr, _ = plan.result(True)
    v2, pr2, conv2 = cg.pagerank(handle, g, None, None, v, pr, 0.85, 1e-5, 500, False, fail_on_nonconvergence=False)
    assert conv2
    np.testing.assert_allclose(pr2.cpu().numpy(), pr.cpu().numpy(), rtol=1e-4)
    # precomputed out-weight sums = out-degrees give the same answer
    outdeg = np.bincount(s, minlength=nv).astype(np.float32)
    v3, pr3, _ = cg.pagerank(handle, g, T(np.arange(nv), np.int32), T(outdeg), None, None, 0.85, 0.0, 10, False, fail_on_nonconvergence=False)
    v4, pr4, _ = cg.pagerank(handle, g, None, None, None, None, 0.85, 0.0, 10, False, fail_on_nonconvergence=False)
    assert np.array_equal(by_vertex(v3, pr3)[0], by_vertex(v4, pr4)[0])


def test_pagerank_is_reproducible(cg, handle, orc, monkeypatch):
    """Every fp32 kernel is bit-reproducible run to run: the single-pass kernels reduce in a fixed order, the tiled default
    accumulates the per-run partials in 64-bit fixed point (integer LDS atomics are order-independent)."""
    s, d = rmat_graph(orc, 15)
    g = make_graph(cg, handle, s, d, None, transposed=True, renumber=True)
    run = lambda: cg.pagerank(handle, g, None, None, None, None, 0.85, 0.0, 10, False, fail_on_nonconvergence=False)[1].cpu().numpy()
    for kern in ("tiled", "flat"):
        monkeypatch.setenv("CUGRAPH_AMD_PAGERANK_KERNEL", kern)
        a, b = run(), run()
        assert np.array_equal(a, b), kern


@pytest.mark.parametrize("weighted", [False, True])
def test_pagerank_plan_tune_keeps_the_bits(cg, handle, orc, weighted):
    """cugraph_amd_pagerank_plan_tune (the plan times itself on several placements of its streamed arrays and keeps the fastest): only addresses
    change -- a tuned plan and an untuned one must produce bit-identical vectors, the tuned placement (cached on the graph) must serve a later
    plan and the ordinary entry point as well, and tuning after the first step is refused."""
    s, d = rmat_graph(orc, 16)
    w = int_weights(s.size) if weighted else None
    g = make_graph(cg, handle, s, d, w, transposed=True, renumber=True, vertices=np.arange(1 << 16))
    a = cg.PageRankPlan(handle, g, 0.85)
    a.step(9)
    _, pa, _ = a.result()
    b = cg.PageRankPlan(handle, g, 0.85)
    assert b.tune(5) > 0.0
    b.step(9)
    vb, pb, _ = b.result()
    assert np.array_equal(pa.cpu().numpy(), pb.cpu().numpy())
    with pytest.raises(Exception):
        b.tune(2)
    c = cg.PageRankPlan(handle, g, 0.85)  # a later plan of the graph: the arrays the tuned plan kept
    c.step(9)
    assert np.array_equal(c.result()[1].cpu().numpy(), pa.cpu().numpy())
    v, pr, _ = cg.pagerank(handle, g, None, None, None, None, 0.85, 0.0, 9, False, fail_on_nonconvergence=False)
    assert np.array_equal(pr.cpu().numpy(), pa.cpu().numpy()) and np.array_equal(v.cpu().numpy(), vb.cpu().numpy())


@pytest.mark.parametrize("hot", [0, 256, 1024, 16384, 32768])
def test_pagerank_lds_tile_sizes_agree(cg, handle, orc, hot):
    s, d = rmat_graph(orc, 15, seed=2)
    nv = 1 << 15
    g = make_graph(cg, handle, s, d, None, transposed=True, renumber=True, vertices=np.arange(nv))
    prev = handle.set_pagerank_hot_tile(hot)
    try:
        v, pr, _ = cg.pagerank(handle, g, None, None, None, None, 0.85, 0.0, 8, False, fail_on_nonconvergence=False)
    finally:
        handle.set_pagerank_hot_tile(prev)
    off, idx, _ = orc.coo_to_cs(nv, d, s)
    truth, _, _ = orc.pagerank(nv, off, idx, None, 0.85, 0.0, 8, acc64=True)
    assert np.max(np.abs(by_vertex(v, pr)[0] - truth)) <= 1e-6


@pytest.mark.parametrize("scale,renumber,transposed", [(12, False, False), (14, True, False), (16, True, True), (18, True, False)])
def test_bfs_rmat_vs_oracle(cg, handle, orc, scale, renumber, transposed):
    s, d = rmat_graph(orc, scale)
    nv = 1 << scale
    g = make_graph(cg, handle, s, d, None, transposed=transposed, renumber=renumber, vertices=np.arange(nv))
    off, idx, _ = orc.coo_to_cs(nv, s, d)
    outdeg = np.diff(off)
    srcs = [int(x) for x in np.nonzero(outdeg > 0)[0][[0, 7, 100]]]
    for src in srcs:
        dist, pred, v = cg.bfs(handle, g, T([src], np.int32), False, 0, True, False)
        verts = v
        dist, pred = by_vertex(v, dist, pred)
        od, _ = orc.bfs(nv, off, idx, [src])
        assert np.array_equal(dist, od)                                  # distances bit-exact
        assert np.array_equal(pred, bfs_expected_parents(s, d, od, verts))  # deterministic parents: smallest internal id one level up
        if not renumber:  # identity numbering: that is the oracle's minimum-id parent
            assert np.array_equal(pred, orc.bfs_min_pred(nv, off, idx, od))
        st = handle.last_traversal_stats()
        assert st["vertices_reached"] == int((od != orc.INT32_MAX).sum())
        assert st["edges_of_reached"] == int(outdeg[od != orc.INT32_MAX].sum())  # what TEPS is scored on (bottom-up levels inspect fewer)
    # depth limit and multi-source
    dist, pred, v = cg.bfs(handle, g, T(srcs, np.int32), False, 2, False, False)
    assert pred.numel() == 0
    (dist,) = by_vertex(v, dist)
    od, _ = orc.bfs(nv, off, idx, srcs, 2)
    assert np.array_equal(dist, od)


@pytest.mark.parametrize("scale,kind,dtype", [(12, "unit", np.float32), (14, "int", np.float32), (16, "int", np.float32),
                                              (14, "real", np.float32), (14, "int", np.float64), (18, "unit", np.float32)])
def test_sssp_rmat_vs_oracle(cg, handle, orc, scale, kind, dtype):
    s, d = rmat_graph(orc, scale)
    nv = 1 << scale
    if kind == "unit":
        w = np.ones(s.size, dtype)
    elif kind == "int":
        w = int_weights(s.size).astype(dtype)
    else:
        w = np.random.default_rng(3).random(s.size).astype(dtype) + dtype(0.01)
    g = make_graph(cg, handle, s, d, w, transposed=False, renumber=True, vertices=np.arange(nv), wdtype=dtype)
    off, idx, ww = orc.coo_to_cs(nv, s, d, w)
    src = int(np.nonzero(np.diff(off) > 0)[0][3])
    v, dist, pred = cg.sssp(handle, g, src, float(np.finfo(dtype).max), True, False)
    dist, pred = by_vertex(v, dist, pred)
    od, _ = orc.sssp(nv, off, idx, ww, src)
    assert np.array_equal(dist, od)  # the fixed point is unique: bit-identical to Dijkstra, any weights
    assert np.array_equal(pred, orc.sssp_min_pred(nv, off, idx, ww, src, od))
    if kind == "unit":  # integer hops == BFS distances bit for bit
        bd, _, bv = cg.bfs(handle, g, T([src], np.int32), False, 0, False, False)
        (bd,) = by_vertex(bv, bd)
        reach = bd != orc.INT32_MAX
        assert np.array_equal(dist[reach], bd[reach].astype(dtype)) and np.all(dist[~reach] == np.finfo(dtype).max)
    # cutoff (with predecessors: the packed path honours it too -- a vertex beyond the cutoff has no parent)
    cut = float(np.median(od[od < np.finfo(dtype).max]))
    v, dist, predc = cg.sssp(handle, g, src, cut, True, False)
    dist, predc = by_vertex(v, dist, predc)
    oc, _ = orc.sssp(nv, off, idx, ww, src, cutoff=cut)
    assert np.array_equal(dist, oc)
    assert np.array_equal(predc, orc.sssp_min_pred(nv, off, idx, ww, src, oc))


@pytest.mark.parametrize("scale,kind", [(13, "int"), (15, "real"), (14, "zero")])
def test_sssp_packed_parents_equal_the_sweep(cg, handle, orc, monkeypatch, scale, kind):
    """Round 5: fp32 SSSP with predecessors lowers (distance, external parent id) with one 64-bit atomicMin per successful relaxation -- the
    reference's lexicographic minimum (sssp_impl.cuh:334) -- instead of sweeping the settled edges afterwards.  Both ways must name the same
    parents (smallest external id over the tight in-edges) and the oracle's; "zero": zero-weight edges and cycles of them (the source keeps -1)."""
    s, d = rmat_graph(orc, scale, seed=11)
    nv = 1 << scale
    rng = np.random.default_rng(2)
    if kind == "int":
        w = int_weights(s.size, seed=6).astype(np.float32)
    elif kind == "real":
        w = (rng.random(s.size) + 0.01).astype(np.float32)
    else:
        w = rng.integers(0, 3, s.size).astype(np.float32)  # a third of the edges weigh nothing
    g = make_graph(cg, handle, s, d, w, transposed=False, renumber=True, vertices=np.arange(nv))
    off, idx, ww = orc.coo_to_cs(nv, s, d, w)
    src = int(np.nonzero(np.diff(off) > 0)[0][5])
    res = {}
    for packed in ("1", "0"):
        monkeypatch.setenv("CUGRAPH_AMD_SSSP_PACKED", packed)
        v, dist, pred = cg.sssp(handle, g, src, float(np.finfo(np.float32).max), True, False)
        res[packed] = by_vertex(v, dist, pred)
    od, _ = orc.sssp(nv, off, idx, ww, src)
    for packed in ("1", "0"):
        assert np.array_equal(res[packed][0], od), packed
        assert res[packed][1][src] == -1
    assert np.array_equal(res["1"][1], res["0"][1])  # the same rule evaluated two ways
    if kind != "zero":
        assert np.array_equal(res["1"][1], orc.sssp_min_pred(nv, off, idx, ww, src, od))
    else:  # zero-weight cycles: "tight in-edge" no longer implies "on a shortest path tree"; every named parent must still be a tight in-neighbour
        for packed in ("1", "0"):
            pred = res[packed][1]
            has = pred >= 0
            assert np.array_equal(has, (od < np.finfo(np.float32).max) & (np.arange(nv) != src))
            key = set(zip(s.tolist(), d.tolist(), w.tolist()))
            vs = np.nonzero(has)[0]
            pick = vs[:: max(1, vs.size // 2000)]
            for vtx in pick:
                pu = int(pred[vtx])
                assert any((pu, int(vtx), float(x)) in key and np.float32(od[pu]) + np.float32(x) == od[vtx] for x in (0.0, 1.0, 2.0))


@pytest.mark.parametrize("scale,kind,dtype", [(12, "int", np.float32), (14, "real", np.float32), (16, "int", np.float32), (14, "int", np.float64), (13, "unit", np.float32),
                                              (14, "zero", np.float32), (17, "int", np.float32)])
def test_sssp_distance_filter_vs_oracle(cg, handle, orc, monkeypatch, scale, kind, dtype):
    """Round 6: the L2-resident distance filter of the wide relaxation rounds (one bit per vertex, d[v] < T; relaxations with nd >= T into such a
    vertex are dropped before they probe its distance) forced onto EVERY round (by default only rounds of >= E / 32 relaxations build it): the
    filter is exact for any threshold, so distances stay bit-identical to Dijkstra and the parents canonical -- integer / real / unit weights,
    zero-weight edges (equal-distance parent ties must still resolve to the smallest id), fp32 packed words and fp64, with a cutoff."""
    monkeypatch.setenv("CUGRAPH_AMD_SSSP_FILTER", "force")
    if kind != "zero":
        _sssp_parity(cg, handle, orc, scale, kind, dtype)
        st = handle.last_traversal_stats()
        assert 0 < st["probes"] <= st["edges_inspected"]
        return
    # zero-weight cycles: a tight in-edge need not lie on a shortest-path tree, so the oracle's parents are not the reference point;
    # the filtered run must name exactly the parents of the unfiltered one (same lexicographic minimum, fewer probes)
    s, d = rmat_graph(orc, scale)
    nv = 1 << scale
    w = (int_weights(s.size) % 3).astype(dtype)
    g = make_graph(cg, handle, s, d, w, transposed=False, renumber=True, vertices=np.arange(nv), wdtype=dtype)
    off, idx, ww = orc.coo_to_cs(nv, s, d, w)
    src = int(np.nonzero(np.diff(off) > 0)[0][3])
    od, _ = orc.sssp(nv, off, idx, ww, src)
    res = {}
    for mode in ("force", "0"):
        monkeypatch.setenv("CUGRAPH_AMD_SSSP_FILTER", mode)
        v, dist, pred = cg.sssp(handle, g, src, float(np.finfo(dtype).max), True, False)
        res[mode] = by_vertex(v, dist, pred)
        assert np.array_equal(res[mode][0], od), mode
    assert np.array_equal(res["force"][1], res["0"][1])


def _sssp_parity(cg, handle, orc, scale, kind, dtype):
    s, d = rmat_graph(orc, scale)
    nv = 1 << scale
    if kind == "unit":
        w = np.ones(s.size, dtype)
    elif kind == "int":
        w = int_weights(s.size).astype(dtype)
    elif kind == "zero":  # a third of the edges weigh nothing: many equal-distance parent candidates
        w = (int_weights(s.size) % 3).astype(dtype)
    else:
        w = np.random.default_rng(3).random(s.size).astype(dtype) + dtype(0.01)
    g = make_graph(cg, handle, s, d, w, transposed=False, renumber=True, vertices=np.arange(nv), wdtype=dtype)
    off, idx, ww = orc.coo_to_cs(nv, s, d, w)
    src = int(np.nonzero(np.diff(off) > 0)[0][3])
    v, dist, pred = cg.sssp(handle, g, src, float(np.finfo(dtype).max), True, False)
    dist, pred = by_vertex(v, dist, pred)
    od, _ = orc.sssp(nv, off, idx, ww, src)
    assert np.array_equal(dist, od)
    assert np.array_equal(pred, orc.sssp_min_pred(nv, off, idx, ww, src, od))
    cut = float(np.median(od[od < np.finfo(dtype).max]))
    v, dist, _ = cg.sssp(handle, g, src, cut, False, False)
    (dist,) = by_vertex(v, dist)
    oc, _ = orc.sssp(nv, off, idx, ww, src, cutoff=cut)
    assert np.array_equal(dist, oc)


def test_csr_input_and_orientation_flip_share_one_numbering(cg, handle, orc):
    s, d = rmat_graph(orc, 12, seed=9)
    nv = 1 << 12
    off, idx, _ = orc.coo_to_cs(nv, s, d)
    props = cg.GraphProperties()
    g = cg.SGGraph(handle, props, T(off, np.int32), T(idx, np.int32), T(np.ones(idx.size), np.float32), store_transposed=False, renumber=True,
                   input_array_format="CSR")
    v1, pr, _ = cg.pagerank(handle, g, None, None, None, None, 0.85, 0.0, 5, False, fail_on_nonconvergence=False)  # builds CSC lazily
    dist, _, v2 = cg.bfs(handle, g, T([0], np.int32), False, 0, False, False)
    assert np.array_equal(v1.cpu().numpy(), v2.cpu().numpy())  # unlike the reference, no re-numbering on the flip
    od, _ = orc.bfs(nv, off, idx, [0])
    assert np.array_equal(by_vertex(v2, dist)[0], od)


@pytest.mark.parametrize("scale,weighted", [(10, False), (13, True), (17, False)])
def test_pagerank_flat_and_row_kernels_agree(cg, handle, orc, scale, weighted, monkeypatch):
    """The edge-balanced kernel (degree-sorted ids) and the row-classed kernel (any numbering) are two
    schedules of the same sums: both must match the oracle, and each other to fp32 round-off."""
    s, d = rmat_graph(orc, scale, seed=4)
    nv = 1 << scale
    w = int_weights(s.size) if weighted else None
    g = make_graph(cg, handle, s, d, w, transposed=True, renumber=True, vertices=np.arange(nv))
    res = {}
    for kern in ("tiled", "flat", "rows"):
        monkeypatch.setenv("CUGRAPH_AMD_PAGERANK_KERNEL", kern)
        v, pr, _ = cg.pagerank(handle, g, None, None, None, None, 0.85, 0.0, 12, False, fail_on_nonconvergence=False)
        res[kern] = by_vertex(v, pr)[0]
    off, idx, ww = orc.coo_to_cs(nv, d, s, w)
    truth, _, _ = orc.pagerank(nv, off, idx, ww, 0.85, 0.0, 12, acc64=True)
    for kern in res:
        assert np.max(np.abs(res[kern] - truth)) <= 1e-6
        assert np.max(np.abs(res[kern] - truth) / truth) <= 2e-5, kern
    np.testing.assert_allclose(res["flat"], res["rows"], rtol=1e-5)
    np.testing.assert_allclose(res["tiled"], res["rows"], rtol=1e-5)


@pytest.mark.parametrize("kern,tile", [("tiled", 0), ("tiled", 256), ("tiled", 1000), ("flat", 0)])
def test_pagerank_ragged_ranges(cg, handle, orc, kern, tile, monkeypatch):
    """Edge counts that are not multiples of the per-wavefront chunk, a hub row spanning many wavefront ranges (and,
    tiled, many source tiles), and single-edge rows: the stitched partial sums must still be exact."""
    monkeypatch.setenv("CUGRAPH_AMD_PAGERANK_KERNEL", kern)
    prev = handle.set_pagerank_hot_tile(tile if tile else -1)
    rng = np.random.default_rng(11)
    nv = 5000
    hub_in = rng.integers(0, nv, 70001)                      # 70001 in-edges of vertex 7
    s = np.concatenate([hub_in, rng.integers(0, nv, 12345), np.arange(100, 1100)])
    d = np.concatenate([np.full(hub_in.size, 7), rng.integers(0, nv, 12345), np.arange(2000, 3000)])
    g = make_graph(cg, handle, s, d, None, transposed=True, renumber=True, vertices=np.arange(nv))
    try:
        v, pr, _ = cg.pagerank(handle, g, None, None, None, None, 0.85, 0.0, 15, False, fail_on_nonconvergence=False)
    finally:
        handle.set_pagerank_hot_tile(prev)
    off, idx, _ = orc.coo_to_cs(nv, d.astype(np.int32), s.astype(np.int32))
    truth, _, _ = orc.pagerank(nv, off, idx, None, 0.85, 0.0, 15, acc64=True)
    got = by_vertex(v, pr)[0]
    assert np.max(np.abs(got - truth) / truth) <= 2e-5


@pytest.mark.parametrize("tile", [0, 512])
def test_pagerank_tiled_fp64_weights_and_personalization(cg, handle, orc, tile):
    """fp64 weights select the double-precision instantiation of both phases; personalization goes through the fused epilogue."""
    scale = 13
    s, d = rmat_graph(orc, scale, seed=9)
    nv = 1 << scale
    w = int_weights(s.size, seed=3).astype(np.float64)
    g = make_graph(cg, handle, s, d, w, transposed=True, renumber=True, vertices=np.arange(nv), wdtype=np.float64)
    prev = handle.set_pagerank_hot_tile(tile if tile else -1)
    try:
        v, pr, _ = cg.pagerank(handle, g, None, None, None, None, 0.85, 0.0, 15, False, fail_on_nonconvergence=False)
        pv = np.array([3, 77, 1000, 4095], np.int32)
        pw = np.array([0.1, 0.2, 0.3, 0.4], np.float64)
        v2, pr2, _ = cg.personalized_pagerank(handle, g, None, None, None, None, T(pv, np.int32), T(pw, np.float64), 0.85, 0.0, 15, False,
                                              fail_on_nonconvergence=False)
    finally:
        handle.set_pagerank_hot_tile(prev)
    off, idx, ww = orc.coo_to_cs(nv, d, s, w)
    truth, _, _ = orc.pagerank(nv, off, idx, ww, 0.85, 0.0, 15, acc64=True, dtype=np.float64)
    got = by_vertex(v, pr)[0]
    assert got.dtype == np.float64
    np.testing.assert_allclose(got, truth, rtol=1e-9)
    truth2, _, _ = orc.pagerank(nv, off, idx, ww, 0.85, 0.0, 15, acc64=True, personalization=(pv, pw), dtype=np.float64)
    np.testing.assert_allclose(by_vertex(v2, pr2)[0], truth2, rtol=1e-9, atol=1e-18)


# ------------------------------------------------------------------ full-size checks through size-independent properties
def _rmat_on_device(cg, handle, scale, weights=None, transposed=False):
    import torch

    nv, ne = 1 << scale, 16 << scale
    src, dst = cg.generate_rmat_edgelist(handle, scale, ne)
    w = None
    if weights == "unit":
        w = torch.ones(ne, dtype=torch.float32, device="cuda")
    elif weights == "int":
        w = torch.randint(1, 256, (ne,), generator=torch.Generator(device="cuda").manual_seed(7), device="cuda").to(torch.float32)
    verts = torch.arange(nv, dtype=torch.int32, device="cuda")
    g = cg.SGGraph(handle, cg.GraphProperties(is_multigraph=True), src, dst, w, store_transposed=transposed, renumber=True, vertices_array=verts)
    return g, src.long(), dst.long(), w, nv, ne


@pytest.mark.parametrize("scale", [22, 24])
def test_bfs_full_size_properties(cg, handle, scale):
    """RMAT-22 / RMAT-24 (BASELINE.json config 3): the distance vector is THE BFS solution iff d[s] = 0, no edge skips a level
    (d[v] <= d[u] + 1 for every edge out of a reached u) and every reached v != s has an in-neighbour one level up; the
    returned parent must be such an in-neighbour.  Checked for all ~2.7e8 edges on the device, both BFS directions."""
    import torch

    g, src, dst, _, nv, ne = _rmat_on_device(cg, handle, scale)
    outdeg = torch.bincount(src, minlength=nv)
    root = int(torch.nonzero(outdeg > 100)[3])
    INF = 2147483647
    res = {}
    for call in range(2):  # the second call on a directed graph runs direction-optimised (CSC built), the first push-only
        d, p, v = cg.bfs(handle, g, torch.tensor([root], dtype=torch.int32, device="cuda"), False, 0, True, False)
        dist = torch.empty(nv, dtype=torch.int64, device="cuda"); dist[v.long()] = d.long()
        pred = torch.empty(nv, dtype=torch.int64, device="cuda"); pred[v.long()] = p.long()
        assert int(dist[root]) == 0 and int(pred[root]) == -1
        du, dv = dist[src], dist[dst]
        reached_u = du != INF
        assert bool((dv[reached_u] <= du[reached_u] + 1).all())                       # no edge skips a level (and reaches its head)
        has_parent = torch.zeros(nv, dtype=torch.bool, device="cuda")
        has_parent[dst[reached_u & (dv == du + 1)]] = True
        reached = dist != INF
        nonroot = reached.clone(); nonroot[root] = False
        assert bool(has_parent[nonroot].all())                                         # every level is supported from the one above
        assert bool((pred[~reached] == -1).all()) and bool((pred[nonroot] >= 0).all())
        assert bool((dist[pred[nonroot]] == dist[nonroot] - 1).all())                  # parents sit one level up ...
        key = src * nv + dst
        pk = pred[nonroot] * nv + torch.nonzero(nonroot).flatten()
        assert bool(torch.isin(pk, key).all())                                         # ... and are joined to their child by an edge
        res[call] = (dist, pred)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])     # push-only == direction-optimised, bit for bit


def test_sssp_full_size_properties(cg, handle):
    """RMAT-22, integer weights (exact fp32 sums): d is THE shortest-path solution iff d[s] = 0, every edge is relaxed
    (d[v] <= d[u] + w) and every reached v != s has a tight in-edge; with unit weights it must equal the BFS levels."""
    import torch

    for kind in ("int", "unit"):
        g, src, dst, w, nv, ne = _rmat_on_device(cg, handle, 22, weights=kind)
        outdeg = torch.bincount(src, minlength=nv)
        root = int(torch.nonzero(outdeg > 100)[5])
        v, d, p = cg.sssp(handle, g, root, 3.0e38, True, False)
        dist = torch.empty(nv, dtype=torch.float32, device="cuda"); dist[v.long()] = d
        pred = torch.empty(nv, dtype=torch.int64, device="cuda"); pred[v.long()] = p.long()
        FMAX = torch.finfo(torch.float32).max
        assert float(dist[root]) == 0.0
        du, dv = dist[src], dist[dst]
        ru = du != FMAX
        assert bool((dv[ru] <= du[ru] + w[ru]).all())
        tight = torch.zeros(nv, dtype=torch.bool, device="cuda")
        tight[dst[ru & (dv == du + w)]] = True
        reached = dist != FMAX
        nonroot = reached.clone(); nonroot[root] = False
        assert bool(tight[nonroot].all())
        assert bool((pred[nonroot] >= 0).all()) and bool((dist[pred[nonroot]] < dist[nonroot]).all())  # (weights >= 1: a parent is strictly closer)
        # the parent is THE canonical one: the smallest id among the in-neighbours joined by a tight edge, d[u] + w(u, v) == d[v] (the reference's
        # lexicographic minimum over (distance, predecessor), sssp_impl.cuh:334) -- which also proves that (pred[v], v) is an edge and that it is tight
        te = ru & (dv == du + w)
        want = torch.full((nv,), nv, dtype=torch.int64, device="cuda")
        want.scatter_reduce_(0, dst[te].long(), src[te].long(), reduce="amin", include_self=True)
        assert torch.equal(pred[nonroot], want[nonroot])
        assert int(pred[root]) == -1 and bool((pred[~reached] == -1).all())
        if kind == "unit":
            bd, _, bv = cg.bfs(handle, g, torch.tensor([root], dtype=torch.int32, device="cuda"), False, 0, False, False)
            lev = torch.empty(nv, dtype=torch.int64, device="cuda"); lev[bv.long()] = bd.long()
            assert torch.equal(lev[reached].to(torch.float32), dist[reached]) and bool((lev[~reached] == 2147483647).all())


def test_pagerank_full_size_properties(cg, handle, monkeypatch):
    """RMAT-22 (BASELINE.json config 2): mass is conserved, two independent kernels (column-tiled two-phase vs single-pass
    gather) agree to fp32 round-off after 20 iterations, and one more iteration of either reproduces the other's update."""
    import torch

    g, src, dst, _, nv, ne = _rmat_on_device(cg, handle, 22, transposed=True)
    out = {}
    for kern in ("tiled", "flat"):
        monkeypatch.setenv("CUGRAPH_AMD_PAGERANK_KERNEL", kern)
        v, pr, _ = cg.pagerank(handle, g, None, None, None, None, 0.85, 0.0, 20, False, fail_on_nonconvergence=False)
        x = torch.empty(nv, dtype=torch.float32, device="cuda"); x[v.long()] = pr
        out[kern] = x
        assert abs(float(x.double().sum()) - 1.0) < 1e-5
    a, b = out["tiled"], out["flat"]
    assert float((a - b).abs().max()) <= 1e-8 and float(((a - b).abs() / b).max()) <= 2e-5
    # one explicit power iteration in fp64 from the 19-iteration state reproduces the 20-iteration state
    monkeypatch.setenv("CUGRAPH_AMD_PAGERANK_KERNEL", "tiled")
    v, pr19, _ = cg.pagerank(handle, g, None, None, None, None, 0.85, 0.0, 19, False, fail_on_nonconvergence=False)
    p19 = torch.empty(nv, dtype=torch.float64, device="cuda"); p19[v.long()] = pr19.double()
    outw = torch.bincount(src, minlength=nv).double()
    xs = p19 / torch.where(outw == 0, torch.ones_like(outw), outw)
    y = torch.zeros(nv, dtype=torch.float64, device="cuda").index_add_(0, dst, xs[src] * 0.85)
    dangling = p19[outw == 0].sum()
    expect = y + (0.85 * dangling + 0.15) / nv
    assert float(((a.double() - expect).abs() / expect).max()) <= 5e-6


def test_pagerank_config2_rmat22_vs_oracle(cg, handle, orc, monkeypatch):
    """BASELINE.json config 2 at full size: RMAT-22 (67 M edges) PageRank fp32, 20 fixed iterations, HIP path vs the CPU
    oracle (restatement of pagerank_reference / detail::pagerank, fp64 accumulation): max|delta| <= 1e-6 AND relative <= 2e-5 on
    every vertex (reference tolerance: cpp/tests/link_analysis/pagerank_test.cpp:328-334, 1e-3 relative).  Run twice: with the
    rows without in-edges left out of the per-iteration epilogue (default) and with every row visited."""
    scale, iters = 22, 20
    nv, ne = 1 << scale, 16 << scale
    s, d = orc.rmat(scale, ne)
    off, idx, _ = orc.coo_to_cs(nv, d, s)
    truth, it, _ = orc.pagerank(nv, off, idx, None, 0.85, 0.0, iters, acc64=True)
    assert it == iters
    g = make_graph(cg, handle, s, d, None, transposed=True, renumber=True, vertices=np.arange(nv))
    got = {}
    for mode in ("const_rows", "all_rows"):
        if mode == "all_rows":
            monkeypatch.setenv("CUGRAPH_AMD_PAGERANK_ALL_ROWS", "1")
        v, pr, conv = cg.pagerank(handle, g, None, None, None, None, 0.85, 0.0, iters, False, fail_on_nonconvergence=False)
        (pr,) = by_vertex(v, pr)
        got[mode] = pr
        assert np.max(np.abs(pr - truth)) <= 1e-6
        rel = np.max(np.abs(pr - truth) / np.maximum(truth, 1e-30))
        assert rel <= 2e-5, (mode, rel)
        assert abs(float(pr.astype(np.float64).sum()) - 1.0) < 1e-4
    assert np.max(np.abs(got["const_rows"] - got["all_rows"]) / got["all_rows"]) <= 1e-6
    # the stopping rule sees the same L1 change either way: same iteration count to epsilon
    monkeypatch.delenv("CUGRAPH_AMD_PAGERANK_ALL_ROWS", raising=False)
    p1 = cg.PageRankPlan(handle, g, 0.85); n1, c1 = p1.step(100, epsilon=1e-7)
    monkeypatch.setenv("CUGRAPH_AMD_PAGERANK_ALL_ROWS", "1")
    p2 = cg.PageRankPlan(handle, g, 0.85); n2, c2 = p2.step(100, epsilon=1e-7)
    tr, it2, conv2 = orc.pagerank(nv, off, idx, None, 0.85, 1e-7, 100, acc64=True)
    assert c1 and c2 and conv2 and n1 == n2 and abs(n1 - it2) <= 1, (n1, n2, it2)


def test_bfs_sssp_config3_rmat24_vs_oracle(cg, handle, orc):
    """BASELINE.json config 3 at full size: RMAT-24 (268 M edges) BFS from 3 roots and SSSP (unit weights = integer hops, and
    integer weights 1..255) from 1 root: distances bit-exact vs the CPU oracle (bfs_reference / Dijkstra sssp_reference
    restatements; cpp/tests/traversal/bfs_test.cpp:213-233, sssp_test.cpp:212-240)."""
    import torch

    scale = 24
    nv, ne = 1 << scale, 16 << scale
    s, d = orc.rmat(scale, ne)
    rng = np.random.default_rng(7)
    wint = rng.integers(1, 256, ne).astype(np.float32)
    coff, cidx, cw = orc.coo_to_cs(nv, s, d, wint)
    outdeg = np.diff(coff)
    roots = [int(r) for r in np.flatnonzero(outdeg > 50)[[3, 1000, 50000]]]
    g = make_graph(cg, handle, s, d, wint, transposed=False, renumber=True, vertices=np.arange(nv))
    for k, root in enumerate(roots):
        dist, pred, bv = cg.bfs(handle, g, T([root], np.int32), False, 0, True, False)
        od, _ = orc.bfs(nv, coff, cidx, [root])
        got_d, got_p = by_vertex(bv, dist, pred)
        assert np.array_equal(got_d, od), f"BFS distances differ (root {root})"
        reached = od != 2147483647
        par = got_p[reached & (np.arange(nv) != root)]
        assert (par >= 0).all() and np.array_equal(od[par] + 1, od[reached & (np.arange(nv) != root)])  # a valid parent one level up
    root = roots[0]
    sv, sd, sp = cg.sssp(handle, g, root, 3.0e38, True, False)
    osd, _ = orc.sssp(nv, coff, cidx, cw, root)
    got_s, got_sp = by_vertex(sv, sd, sp)
    assert np.array_equal(got_s, osd), "SSSP (integer weights) distances differ"
    # the parents of the packed 64-bit relaxation at the configuration's full size: the oracle's canonical ones (smallest id over the tight in-edges)
    assert np.array_equal(got_sp, orc.sssp_min_pred(nv, coff, cidx, cw, root, osd)), "SSSP (integer weights) predecessors differ"
    del g
    ones = np.ones(ne, np.float32)
    g1 = make_graph(cg, handle, s, d, ones, transposed=False, renumber=True, vertices=np.arange(nv))
    sv, sd, _ = cg.sssp(handle, g1, root, 3.0e38, False, False)
    (got_u,) = by_vertex(sv, sd)
    od, _ = orc.bfs(nv, coff, cidx, [root])
    hops = np.where(od == 2147483647, np.float32(np.finfo(np.float32).max), od.astype(np.float32))
    assert np.array_equal(got_u, hops), "SSSP with unit weights must equal the BFS levels bit for bit"


@pytest.mark.parametrize("flags", [dict(drop_self_loops=True), dict(drop_multi_edges=True), dict(symmetrize=True),
                                   dict(drop_self_loops=True, drop_multi_edges=True, symmetrize=True)])
@pytest.mark.parametrize("weighted", [False, True])
def test_graph_creation_flags(cg, handle, orc, flags, weighted):
    """drop_self_loops / drop_multi_edges / symmetrize (graph_sg.cpp:185-248): the graph built with the flags must behave
    exactly like a graph built from the oracle's preprocessed edge list -- same number of edges, same PageRank, same BFS."""
    from cugraph_amd import _capi

    rng = np.random.default_rng(5)
    nv, ne = 300, 6000                                     # dense enough for many multi-edges, reciprocal pairs and self-loops
    s = rng.integers(0, nv, ne).astype(np.int32)
    d = rng.integers(0, nv, ne).astype(np.int32)
    d[:200] = s[:200]                                      # self-loops
    w = rng.integers(1, 9, ne).astype(np.float32) if weighted else None
    es, ed, ew = s, d, w
    if flags.get("drop_self_loops"):
        es, ed, ew = orc.remove_self_loops(es, ed, ew)
    if flags.get("drop_multi_edges"):
        es, ed, ew = orc.remove_multi_edges(es, ed, ew)
    if flags.get("symmetrize"):
        es, ed, ew = orc.symmetrize_edgelist(es, ed, ew)
    props = cg.GraphProperties(is_symmetric=bool(flags.get("symmetrize")), is_multigraph=not flags.get("drop_multi_edges"))
    g = cg.SGGraph(handle, props, T(s, np.int32), T(d, np.int32), None if w is None else T(w, np.float32), store_transposed=True,
                   renumber=True, vertices_array=T(np.arange(nv), np.int32), **flags)
    assert _capi.lib().cugraph_amd_graph_num_edges(g.c_graph_ptr) == es.size
    v, pr, _ = cg.pagerank(handle, g, None, None, None, None, 0.85, 0.0, 30, False, fail_on_nonconvergence=False)
    off, idx, ww = orc.coo_to_cs(nv, ed, es, ew)
    truth, _, _ = orc.pagerank(nv, off, idx, ww, 0.85, 0.0, 30, acc64=True)
    np.testing.assert_allclose(by_vertex(v, pr)[0], truth, rtol=3e-5)
    coff, cidx, _ = orc.coo_to_cs(nv, es, ed)
    dist, _, bv = cg.bfs(handle, g, T([int(es[0])], np.int32), False, 0, False, False)
    od, _ = orc.bfs(nv, coff, cidx, [int(es[0])])
    assert np.array_equal(by_vertex(bv, dist)[0], od)


def test_symmetrize_needs_symmetric_property(cg, handle):
    with pytest.raises(Exception):
        cg.SGGraph(handle, cg.GraphProperties(is_symmetric=False), T([0, 1], np.int32), T([1, 2], np.int32), None, symmetrize=True)


def test_graph_creation_expensive_check(cg, handle):
    """do_expensive_check = TRUE: cpp/tests/c_api/create_graph_test.c:435-535 (symmetric property on an asymmetric edge
    list must fail) and the other checks of create_graph_from_edgelist_impl.cuh:72-126, 1466-1494."""
    src = T([0, 1, 1, 2, 2, 2, 3, 4], np.int32)
    dst = T([1, 3, 4, 0, 1, 3, 5, 5], np.int32)
    wgt = T([0.1, 2.1, 1.1, 5.1, 3.1, 4.1, 7.2, 3.2], np.float32)
    sym = cg.GraphProperties(is_symmetric=True, is_multigraph=False)
    with pytest.raises(ValueError, match="not symmetric"):
        cg.SGGraph(handle, sym, src, dst, wgt, do_expensive_check=True)
    cg.SGGraph(handle, sym, src, dst, wgt, do_expensive_check=False)  # unchecked, as in the reference
    cg.SGGraph(handle, sym, src, dst, wgt, symmetrize=True, do_expensive_check=True)  # symmetrised first, then checked
    import torch
    s2, d2 = torch.cat([src, dst]), torch.cat([dst, src])
    cg.SGGraph(handle, sym, s2, d2, torch.cat([wgt, wgt]), renumber=True, do_expensive_check=True)
    # a symmetric multiset needs equal multiplicities in both directions
    with pytest.raises(ValueError, match="not symmetric"):
        cg.SGGraph(handle, cg.GraphProperties(is_symmetric=True, is_multigraph=True), T([0, 0, 1], np.int32), T([1, 1, 0], np.int32),
                   do_expensive_check=True)
    plain = cg.GraphProperties(is_symmetric=False, is_multigraph=False)
    with pytest.raises(ValueError, match="parallel edges"):
        cg.SGGraph(handle, plain, T([0, 0, 1], np.int32), T([1, 1, 2], np.int32), do_expensive_check=True)
    cg.SGGraph(handle, cg.GraphProperties(is_symmetric=False, is_multigraph=True), T([0, 0, 1], np.int32), T([1, 1, 2], np.int32),
               do_expensive_check=True)
    cg.SGGraph(handle, plain, T([0, 0, 1], np.int32), T([1, 1, 2], np.int32), drop_multi_edges=True, do_expensive_check=True)
    with pytest.raises(ValueError, match="duplicates"):
        cg.SGGraph(handle, plain, T([0, 1], np.int32), T([1, 2], np.int32), vertices_array=T([0, 1, 2, 2], np.int32),
                   do_expensive_check=True)
    with pytest.raises(ValueError, match="consecutive"):
        cg.SGGraph(handle, plain, T([0, 1], np.int32), T([1, 2], np.int32), vertices_array=T([0, 1, 2, 4], np.int32),
                   do_expensive_check=True)
    cg.SGGraph(handle, plain, T([0, 1], np.int32), T([1, 2], np.int32), vertices_array=T([3, 1, 2, 0], np.int32),
               do_expensive_check=True)
    cg.SGGraph(handle, plain, T([10, 11], np.int32), T([11, 12], np.int32), vertices_array=T([10, 11, 12, 40], np.int32), renumber=True,
               do_expensive_check=True)


# ---------------------------------------------------------------- degrees / extract_paths
GOLD_SRC = [0, 1, 1, 2, 2, 2, 3, 4]
GOLD_DST = [1, 3, 4, 0, 1, 3, 5, 5]
GOLD_WGT = [0.1, 2.1, 1.1, 5.1, 3.1, 4.1, 7.2, 3.2]


@pytest.mark.parametrize("store_transposed", [False, True])
@pytest.mark.parametrize("renumber", [False, True])
def test_capi_degrees_golden(cg, handle, store_transposed, renumber):
    """cpp/tests/c_api/degrees_test.c: test_degrees, test_in_degrees, test_out_degrees, test_degrees_subset."""
    g = cg.SGGraph(handle, cg.GraphProperties(), T(GOLD_SRC, np.int32), T(GOLD_DST, np.int32), T(GOLD_WGT, np.float32),
                   store_transposed=store_transposed, renumber=renumber)
    want_in, want_out = np.array([1, 2, 0, 2, 1, 2]), np.array([1, 2, 3, 1, 1, 0])
    v, din, dout = cg.degrees(handle, g)
    v = v.cpu().numpy()
    assert sorted(v.tolist()) == list(range(6))
    assert np.array_equal(din.cpu().numpy(), want_in[v]) and np.array_equal(dout.cpu().numpy(), want_out[v])
    v, din = cg.in_degrees(handle, g)
    assert np.array_equal(din.cpu().numpy(), want_in[v.cpu().numpy()])
    v, dout = cg.out_degrees(handle, g)
    assert np.array_equal(dout.cpu().numpy(), want_out[v.cpu().numpy()])
    v, din, dout = cg.degrees(handle, g, T([2, 3, 5], np.int32))
    assert v.cpu().numpy().tolist() == [2, 3, 5]
    assert din.cpu().numpy().tolist() == [0, 2, 2] and dout.cpu().numpy().tolist() == [3, 1, 0]
    with pytest.raises(ValueError):
        cg.degrees(handle, g, T([2, 17], np.int32))


def test_capi_degrees_symmetric_golden(cg, handle):
    """degrees_test.c test_degrees_symmetric: the out-degrees of a symmetric graph are served from its in-degrees."""
    s = GOLD_SRC + GOLD_DST
    d = GOLD_DST + GOLD_SRC
    g = cg.SGGraph(handle, cg.GraphProperties(is_symmetric=True), T(s, np.int32), T(d, np.int32), T(GOLD_WGT + GOLD_WGT, np.float32), renumber=True)
    v, din, dout = cg.degrees(handle, g)
    want = np.array([2, 4, 3, 3, 2, 2])
    v = v.cpu().numpy()
    assert np.array_equal(din.cpu().numpy(), want[v]) and np.array_equal(dout.cpu().numpy(), want[v])


def test_degrees_rmat_with_multi_edges(cg, handle, orc):
    """every edge counts (multi-edges, self-loops), whichever orientation the graph stores"""
    s, d = rmat_graph(orc, 12)
    nv = 1 << 12
    for st in (False, True):
        g = cg.SGGraph(handle, cg.GraphProperties(is_multigraph=True), T(s, np.int32), T(d, np.int32), store_transposed=st, renumber=True,
                       vertices_array=T(np.arange(nv), np.int32))
        v, din, dout = cg.degrees(handle, g)
        v = v.cpu().numpy()
        assert np.array_equal(din.cpu().numpy(), np.bincount(d, minlength=nv)[v])
        assert np.array_equal(dout.cpu().numpy(), np.bincount(s, minlength=nv)[v])


def test_degrees_partitioned_histogram(cg, handle, orc, monkeypatch):
    """the large-input path of histogram_i32_mapped (keys grouped by their top 16 bits, windowed LDS counters; prims.hip) forced on
    a graph small enough to check: vertex ids spread over a range of 3 * 2^17 so a chunk's window does not cover everything;
    renumbering by degree (the mapped variant) and cugraph_degrees (the plain variant) both go through it"""
    monkeypatch.setenv("CUGRAPH_AMD_HISTOGRAM", "partition")
    scale = 17
    s, d = rmat_graph(orc, scale)
    s, d = s.astype(np.int64) * 3 + 5, d.astype(np.int64) * 3 + 5
    verts = np.arange(1 << scale, dtype=np.int64) * 3 + 5
    for st in (False, True):
        g = cg.SGGraph(handle, cg.GraphProperties(is_multigraph=True), T(s, np.int32), T(d, np.int32), store_transposed=st, renumber=True,
                       vertices_array=T(verts, np.int32))
        v, din, dout = cg.degrees(handle, g)
        v = v.cpu().numpy()
        assert np.array_equal(np.sort(v), verts)
        assert np.array_equal(din.cpu().numpy(), np.bincount(d, minlength=int(verts[-1]) + 1)[v])
        assert np.array_equal(dout.cpu().numpy(), np.bincount(s, minlength=int(verts[-1]) + 1)[v])
        major = din if st else dout  # internal ids descend in the major degree
        assert np.all(np.diff(major.cpu().numpy().astype(np.int64)) <= 0)


@pytest.mark.parametrize("store_transposed", [False, True])
def test_capi_extract_paths_golden(cg, handle, store_transposed):
    """cpp/tests/c_api/extract_paths_test.c: test_bfs_with_extract_paths(_with_transpose): seeds {0}, destinations {5} ->
    max path length 4, path 0 1 3 5."""
    g = cg.SGGraph(handle, cg.GraphProperties(), T(GOLD_SRC, np.int32), T(GOLD_DST, np.int32), T(GOLD_WGT, np.float32),
                   store_transposed=store_transposed, renumber=False)
    dist, pred, verts, paths = cg.bfs_extract_paths(handle, g, T([0], np.int32), T([5], np.int32), depth_limit=10)
    assert paths.shape == (1, 4) and paths.cpu().numpy().tolist() == [[0, 1, 3, 5]]
    # several destinations incl. the source itself and an unreachable vertex: rows padded with -1
    dist, pred, verts, paths = cg.bfs_extract_paths(handle, g, T([0], np.int32), T([4, 0, 2, 5], np.int32))
    assert paths.cpu().numpy().tolist() == [[0, 1, 4, -1], [0, -1, -1, -1], [-1, -1, -1, -1], [0, 1, 3, 5]]


def test_extract_paths_rmat(cg, handle, orc):
    """every extracted path is a shortest path: starts at the source, ends at the destination, consecutive vertices are joined
    by an edge and the length equals the BFS distance"""
    scale = 12
    s, d = rmat_graph(orc, scale)
    nv = 1 << scale
    g = cg.SGGraph(handle, cg.GraphProperties(is_multigraph=True), T(s, np.int32), T(d, np.int32), renumber=True, vertices_array=T(np.arange(nv), np.int32))
    src = int(np.flatnonzero(np.bincount(s, minlength=nv) > 0)[7])
    dest = np.random.default_rng(5).choice(nv, 200, replace=False).astype(np.int32)
    dist, pred, verts, paths = cg.bfs_extract_paths(handle, g, T([src], np.int32), T(dest, np.int32))
    dd = np.empty(nv, np.int64)
    dd[verts.cpu().numpy()] = dist.cpu().numpy()
    edges = set(zip(s.tolist(), d.tolist()))
    paths = paths.cpu().numpy()
    INT32_MAX = orc.INT32_MAX
    reach = dd[dest] != INT32_MAX
    assert paths.shape[1] == 1 + int(dd[dest][reach & (dest != src)].max(initial=0))
    for row, t in zip(paths, dest):
        if dd[t] == INT32_MAX:
            assert (row == -1).all()
            continue
        n = int(dd[t]) + 1
        assert row[0] == src and row[n - 1] == t and (row[n:] == -1).all()
        assert all((int(a), int(b)) in edges for a, b in zip(row[:n - 1], row[1:n]))


# ---------------------------------------------------------------- the reference's generator API
def test_capi_generate_rmat_edgelist(cg, handle, orc):
    """cugraph_generate_rmat_edgelist behind cugraph_rng_state_t / cugraph_coo_t: a fresh state with seed s gives the edge list of
    the oracle (and of the benchmark generator); sizes and ranges as cpp/tests/c_api/generate_rmat_test.c checks them."""
    scale, ne = 10, 5000
    src, dst, w = cg.rmat_edgelist(handle, 0, scale, ne)
    os_, od_ = orc.rmat(scale, ne, seed=0)
    assert w is None and np.array_equal(src.cpu().numpy(), os_) and np.array_equal(dst.cpu().numpy(), od_)
    src7, dst7, _ = cg.rmat_edgelist(handle, 7, scale, ne)
    assert not np.array_equal(src7.cpu().numpy(), os_)
    # clip_and_flip keeps the lower triangle (generate_rmat_edgelist.cuh:90-97)
    s, d, _ = cg.rmat_edgelist(handle, 0, scale, ne, clip_and_flip=True)
    s, d = s.cpu().numpy(), d.cpu().numpy()
    assert (s >= d).all() and s.min() >= 0 and s.max() < (1 << scale)
    # the id scramble is one permutation of [0, 2^scale) applied to both endpoints
    s2, d2, _ = cg.rmat_edgelist(handle, 0, scale, ne, scramble_vertex_ids=True)
    s2, d2 = s2.cpu().numpy(), d2.cpu().numpy()
    perm = {}
    for a, b in list(zip(os_.tolist(), s2.tolist())) + list(zip(od_.tolist(), d2.tolist())):
        assert perm.setdefault(a, b) == b
    assert len(set(perm.values())) == len(perm) and max(perm.values()) < (1 << scale) and min(perm.values()) >= 0
    assert any(k != v for k, v in perm.items())
    # weights: uniform in [lo, hi), the requested type
    import torch
    _, _, w = cg.rmat_edgelist(handle, 3, scale, ne, include_edge_weights=True, minimum_weight=2.0, maximum_weight=5.0)
    w = w.cpu().numpy()
    assert w.dtype == np.float32 and w.size == ne and w.min() >= 2.0 and w.max() < 5.0 and 3.3 < w.mean() < 3.7
    _, _, w64 = cg.rmat_edgelist(handle, 3, scale, ne, include_edge_weights=True, dtype=torch.float64)
    assert w64.cpu().numpy().dtype == np.float64
    with pytest.raises(ValueError):
        cg.rmat_edgelist(handle, 0, scale, ne, a=0.9, b=0.2, c=0.1)


# ---------------------------------------------------------------- MatrixMarket reader
def test_read_matrix_market(cg, handle, golden, tmp_path):
    """The karate graph written the way datasets/karate.mtx stores it (symmetric, one triangle, 1-based, comments) reads back as
    the 156 directed edges of karate.csv and reproduces the pylibcugraph PageRank golden; general / pattern / integer files,
    isolated vertices and malformed input."""
    gr = golden["graphs"]["karate.csv"]
    s, d = np.array(gr["src"]), np.array(gr["dst"])
    lower = s > d
    path = tmp_path / "karate.mtx"
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real symmetric\n% Zachary karate club\n%\n34 34 78\n")
        for a, b in zip(s[lower], d[lower]):
            f.write(f"{a + 1} {b + 1} 1.0\n")
    src, dst, w, nv, sym, hw = cg.read_matrix_market(handle, path)
    assert nv == 34 and sym and hw and src.numel() == 156
    got = sorted(zip(src.cpu().numpy().tolist(), dst.cpu().numpy().tolist()))
    assert got == sorted(zip(s.tolist(), d.tolist())) and bool((w == 1.0).all())
    p = golden["pylibcugraph_pagerank"]["params"]
    g = cg.SGGraph(handle, cg.GraphProperties(is_symmetric=True), src, dst, w, store_transposed=True, renumber=False,
                   vertices_array=T(np.arange(nv), np.int32))
    v, pr = cg.pagerank(handle, g, None, None, None, None, p["alpha"], p["epsilon"], p["max_iterations"], False)
    exp = golden["pylibcugraph_pagerank"]["karate.csv"]
    np.testing.assert_allclose(pr.cpu().numpy(), np.array(exp["pagerank"]), rtol=p["rel_tol"], atol=5e-7)
    # general + integer values, a self-loop, an isolated last vertex
    path2 = tmp_path / "small.mtx"
    path2.write_text("%%MatrixMarket matrix coordinate integer general\n5 5 4\n1 2 7\n2 3 2\n3 3 9\n4 1 1\n")
    src, dst, w, nv, sym, hw = cg.read_matrix_market(handle, path2)
    assert (nv, sym, hw) == (5, False, True)
    assert src.cpu().numpy().tolist() == [0, 1, 2, 3] and dst.cpu().numpy().tolist() == [1, 2, 2, 0] and w.cpu().numpy().tolist() == [7.0, 2.0, 9.0, 1.0]
    # pattern + symmetric: weights 1, the diagonal entry is not mirrored
    path3 = tmp_path / "pat.mtx"
    path3.write_text("%%MatrixMarket matrix coordinate pattern symmetric\n3 3 3\n2 1\n3 1\n3 3\n")
    src, dst, w, nv, sym, hw = cg.read_matrix_market(handle, path3)
    assert (nv, sym, hw) == (3, True, False) and src.numel() == 5 and bool((w == 1.0).all())
    assert sorted(zip(src.cpu().numpy().tolist(), dst.cpu().numpy().tolist())) == [(0, 1), (0, 2), (1, 0), (2, 0), (2, 2)]
    for bad in ("garbage\n1 1 0\n", "%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n", "%%MatrixMarket matrix coordinate real general\n3 3 2\n1 2 1.0\n",
                "%%MatrixMarket matrix coordinate real general\n3 3 1\n1 4 1.0\n", "%%MatrixMarket matrix coordinate real general\n3 4 0\n"):
        pb = tmp_path / "bad.mtx"
        pb.write_text(bad)
        with pytest.raises(ValueError):
            cg.read_matrix_market(handle, pb)
    with pytest.raises(ValueError):
        cg.read_matrix_market(handle, tmp_path / "missing.mtx")


# ---------------------------------------------------------------- Louvain
LOUVAIN_SRC = [0, 1, 1, 2, 2, 2, 3, 4, 1, 3, 4, 0, 1, 3, 5, 5]
LOUVAIN_DST = [1, 3, 4, 0, 1, 3, 5, 5, 0, 1, 1, 2, 2, 2, 3, 4]
LOUVAIN_WGT = [0.1, 2.1, 1.1, 5.1, 3.1, 4.1, 7.2, 3.2] * 2


def canonical_partition(c):
    """labels renamed by first occurrence (cluster ids are arbitrary names)"""
    seen = {}
    return [seen.setdefault(int(x), len(seen)) for x in c]


@pytest.mark.parametrize("store_transposed", [False, True])
def test_capi_louvain_goldens(cg, handle, orc, store_transposed):
    """cpp/tests/c_api/louvain_test.c: test_louvain ({0,0,0,1,1,1}, Q = 0.215969) and test_louvain_no_weight ({1,1,1,1,0,0},
    Q = 0.125); max_level 10, threshold 1e-7, resolution 1.0 -- cluster ids included."""
    props = cg.GraphProperties(is_symmetric=True)
    g = cg.SGGraph(handle, props, T(LOUVAIN_SRC, np.int32), T(LOUVAIN_DST, np.int32), T(LOUVAIN_WGT, np.float32), store_transposed=store_transposed,
                   renumber=False)
    v, c, q = cg.louvain(handle, g, 10, 1e-7, 1.0, False)
    (c,) = by_vertex(v, c)
    assert c.tolist() == [0, 0, 0, 1, 1, 1] and nearly_equal(q, 0.215969, 0.001)
    g = cg.SGGraph(handle, props, T(LOUVAIN_SRC, np.int32), T(LOUVAIN_DST, np.int32), None, store_transposed=store_transposed, renumber=False)
    v, c, q = cg.louvain(handle, g, 10, 1e-7, 1.0, False)
    (c,) = by_vertex(v, c)
    assert c.tolist() == [1, 1, 1, 1, 0, 0] and nearly_equal(q, 0.125, 0.001)
    # max_level 1: one sweep level only (the level's clustering, not contracted further)
    v, c1, q1 = cg.louvain(handle, g, 1, 1e-7, 1.0, False)
    oc, oq, _ = orc.louvain(6, LOUVAIN_SRC, LOUVAIN_DST, None, 1, 1e-7, 1.0)
    assert canonical_partition(by_vertex(v, c1)[0]) == canonical_partition(oc) and abs(q1 - oq) <= 1e-12


def test_louvain_reference_karate_goldens(cg, handle, orc, golden):
    """cpp/tests/community/louvain_test.cpp:228-237 -- the reference's own expectations for karate (renumber = false, float weights): modularity
    0.39907956 (defaults, and max_level 20 / threshold 1e-3), 0.48573306 at resolution 0.8; ASSERT_FLOAT_EQ = 4 float ulps.  The second and third
    level of these runs are decided by how graph_contraction NUMBERS the coarse vertices (by degree, descending: coarse_degree_order in
    csrc/louvain.hip); with label-order ids the run ends at 0.4197896.  Clusters equal the restatement's vertex for vertex."""
    k = golden["graphs"]["karate.csv"]
    src, dst = np.array(k["src"], np.int32), np.array(k["dst"], np.int32)
    w = np.ones(src.size, np.float32)
    g = cg.SGGraph(handle, cg.GraphProperties(is_symmetric=True), T(src, np.int32), T(dst, np.int32), T(w, np.float32), renumber=False)
    for args, q_ref in (((100, 1e-7, 1.0), 0.39907956), ((20, 1e-3, 1.0), 0.39907956), ((100, 1e-3, 0.8), 0.48573306)):
        v, c, q = cg.louvain(handle, g, args[0], args[1], args[2], False)
        (c,) = by_vertex(v, c)
        assert abs(np.float32(q) - np.float32(q_ref)) <= 4 * np.spacing(np.float32(q_ref)), (args, q)
        oc, oq, olevels = orc.louvain(34, src, dst, w, *args)
        assert olevels == 3 and np.array_equal(c, oc) and abs(q - oq) <= 1e-12


@pytest.mark.parametrize("scale,resolution", [(8, 1.0), (10, 1.0), (10, 0.5)])
def test_louvain_rmat_vs_oracle(cg, handle, orc, scale, resolution):
    """Undirected RMAT with integer weights (every sum is exact): the clustering must equal the oracle's vertex for vertex, the
    reported modularity must be the modularity of that clustering."""
    s, d = orc.rmat(scale, 8 << scale, seed=5)
    keep = s != d
    lo, hi = np.minimum(s[keep], d[keep]), np.maximum(s[keep], d[keep])
    pairs = np.unique(np.stack([lo, hi], 1), axis=0)
    wt = (1 + (pairs[:, 0] * 7 + pairs[:, 1] * 13) % 8).astype(np.float32)
    src = np.concatenate([pairs[:, 0], pairs[:, 1]]).astype(np.int32)
    dst = np.concatenate([pairs[:, 1], pairs[:, 0]]).astype(np.int32)
    w = np.concatenate([wt, wt])
    nv = 1 << scale
    o = np.lexsort((dst, src))  # the order the graph stores its edges in
    oc, oq, olevels = orc.louvain(nv, src[o], dst[o], w[o], 100, 1e-7, resolution)
    g = cg.SGGraph(handle, cg.GraphProperties(is_symmetric=True), T(src, np.int32), T(dst, np.int32), T(w, np.float32), renumber=False,
                   vertices_array=T(np.arange(nv), np.int32))
    v, c, q = cg.louvain(handle, g, 100, 1e-7, resolution, False)
    (c,) = by_vertex(v, c)
    assert abs(q - orc.louvain_modularity(src, dst, w, c, resolution)) <= 1e-9
    assert abs(q - oq) <= 1e-9
    assert np.array_equal(c, oc)
    assert olevels >= 2 and len(np.unique(c)) < nv // 2


@pytest.mark.parametrize("scale", [10, 12])
def test_louvain_real_weights_vs_oracle(cg, handle, orc, scale):
    """fp32 weights that are not integers: the fixed-point sums of the GPU path (scale 2^s from the total weight) and the sequential
    fp64 sums of the oracle are both exact for fp32 inputs of moderate range, so the clustering is still equal vertex for vertex"""
    s, d = orc.rmat(scale, 8 << scale, seed=7)
    keep = s != d
    pairs = np.unique(np.stack([np.minimum(s[keep], d[keep]), np.maximum(s[keep], d[keep])], 1), axis=0)
    wt = (np.random.default_rng(11).random(len(pairs)) + 0.1).astype(np.float32)
    src = np.concatenate([pairs[:, 0], pairs[:, 1]]).astype(np.int32)
    dst = np.concatenate([pairs[:, 1], pairs[:, 0]]).astype(np.int32)
    w = np.concatenate([wt, wt])
    nv = 1 << scale
    o = np.lexsort((dst, src))
    oc, oq, _ = orc.louvain(nv, src[o], dst[o], w[o], 100, 1e-7, 1.0)
    g = cg.SGGraph(handle, cg.GraphProperties(is_symmetric=True), T(src, np.int32), T(dst, np.int32), T(w, np.float32), renumber=False,
                   vertices_array=T(np.arange(nv), np.int32))
    v, c, q = cg.louvain(handle, g, 100, 1e-7, 1.0, False)
    (c,) = by_vertex(v, c)
    assert np.array_equal(c, oc)
    assert abs(q - oq) <= 1e-9


def louvain_rmat_input(orc, scale, edge_factor=8, seed=5):
    """undirected simple RMAT graph with integer weights 1..8 (every sum exact), both directions listed, sorted by (src, dst)"""
    s, d = orc.rmat(scale, edge_factor << scale, seed=seed)
    keep = s != d
    lo, hi = np.minimum(s[keep], d[keep]).astype(np.int64), np.maximum(s[keep], d[keep]).astype(np.int64)
    key = np.unique(lo << 32 | hi)
    lo, hi = (key >> 32).astype(np.int32), (key & 0xFFFFFFFF).astype(np.int32)
    wt = (1 + (lo.astype(np.int64) * 7 + hi.astype(np.int64) * 13) % 8).astype(np.float32)
    src, dst, w = np.concatenate([lo, hi]), np.concatenate([hi, lo]), np.concatenate([wt, wt])
    o = np.lexsort((dst, src))
    return src[o], dst[o], w[o]


@pytest.mark.parametrize("scale", [16, 18, 20])
def test_louvain_scale_vs_oracle(cg, handle, orc, scale):
    """Louvain beyond toy size (hub rows of 10^4..10^5 edges walked by whole wavefronts, fixed-point cluster / coarse-edge weights,
    chunked reductions): the clustering equals the C oracle's (oracle.c: orc_louvain, itself checked against the numpy
    restatement in tests/test_oracle.py) vertex for vertex, the modularity to 1e-9, level for level the same number of sweeps."""
    src, dst, w = louvain_rmat_input(orc, scale)
    nv = 1 << scale
    oc, oq, olevels, _ = orc.louvain_c(nv, src, dst, w, 100, 1e-7, 1.0)
    g = cg.SGGraph(handle, cg.GraphProperties(is_symmetric=True), T(src, np.int32), T(dst, np.int32), T(w, np.float32), renumber=False,
                   vertices_array=T(np.arange(nv), np.int32))
    v, c, q = cg.louvain(handle, g, 100, 1e-7, 1.0, False)
    (c,) = by_vertex(v, c)
    assert abs(q - oq) <= 1e-9
    assert np.array_equal(c, oc)
    assert abs(q - orc.louvain_modularity(src, dst, w, c, 1.0)) <= 1e-9


@pytest.mark.parametrize("scale,weights", [(14, "int"), (16, "real")])
def test_louvain_hash_path_equals_sorted_path(cg, handle, orc, monkeypatch, scale, weights):
    """Round 3: rows of at most 512 edges take the LDS hash path (k_lv_hash_chunks), hubs a global hash table (k_lv_hub_*) or the
    sorted path.  All accumulate the same fixed-point integers, so the whole run -- clusters, modularity, hierarchy -- must not depend
    on which path a row takes: the default against hubs-sorted (CUGRAPH_AMD_LOUVAIN_HUB=sort) and everything-sorted
    (CUGRAPH_AMD_LOUVAIN_HASH=0), on a graph with hubs, isolated vertices, self-loops and multi-edges, and against the C oracle."""
    src, dst, w = louvain_rmat_input(orc, scale)
    nv = 1 << scale
    rng = np.random.default_rng(3)
    loops = rng.integers(0, nv, 500).astype(np.int32)           # self-loops (cluster_subtract path)
    dup = rng.integers(0, src.size, 2000)                       # repeated edges, both directions
    # two stars, so that there are rows of more than 4096 edges (several LDS work items per row) at these sizes: 9 000 and 5 000 spokes
    hub_a, hub_b = np.full(9000, 7, np.int32), np.full(5000, 11, np.int32)
    spoke_a, spoke_b = rng.permutation(nv)[:9000].astype(np.int32), rng.permutation(nv)[:5000].astype(np.int32)
    star_s, star_d = np.concatenate([hub_a, spoke_a, hub_b, spoke_b]), np.concatenate([spoke_a, hub_a, spoke_b, hub_b])
    src, dst = np.concatenate([src, loops, src[dup], dst[dup], star_s]), np.concatenate([dst, loops, dst[dup], src[dup], star_d])
    wl = np.full(500, 2.0, np.float32)
    w = np.concatenate([w, wl, w[dup], w[dup], np.full(star_s.size, 3.0, np.float32)])
    if weights == "real":
        w = (w * np.float32(0.37) + np.float32(0.25)).astype(np.float32)
    o = np.lexsort((dst, src))
    src, dst, w = src[o], dst[o], w[o]
    res = {}
    # default of round 3 (LDS hash + global hub hash); hubs sorted; everything sorted; round 4: the default (mid rows and big rows in LDS
    # tables too), mid rows back on the sorted path, big rows back on it, and big rows whose table "fills up" after 16 clusters (the fall-back
    # to the sorted path in the middle of a level)
    variants = {"1hash": {"HASH": "1", "HUB": "hash"}, "1sort": {"HASH": "1", "HUB": "sort", "MID": "0"}, "0sort": {"HASH": "0", "HUB": "sort"},
                "default": {}, "mid0": {"MID": "0"}, "big0": {"BIG": "0"}, "overflow": {"BIG_SLOTS": "16"},
                # round 6: the big rows' pairs partitioned by work item every sweep (k_lv_big_partition; by default only where the items would re-read a row many
                # times over: RMAT-26-sized hubs), forced on and forced off, and forced on together with a table that fills up
                "partition": {"PARTITION": "1"}, "no_partition": {"PARTITION": "0"}, "partition_overflow": {"PARTITION": "1", "BIG_SLOTS": "16"}}
    for name, env in variants.items():
        for k in ("HASH", "HUB", "MID", "BIG", "BIG_SLOTS", "PARTITION"):
            monkeypatch.delenv("CUGRAPH_AMD_LOUVAIN_" + k, raising=False)
        for k, val in env.items():
            monkeypatch.setenv("CUGRAPH_AMD_LOUVAIN_" + k, val)
        g = cg.SGGraph(handle, cg.GraphProperties(is_symmetric=True, is_multigraph=True), T(src, np.int32), T(dst, np.int32), T(w, np.float32), renumber=False,
                       vertices_array=T(np.arange(nv), np.int32))
        v, c, q = cg.louvain(handle, g, 100, 1e-7, 1.0, False)
        res[name] = (by_vertex(v, c)[0], q)
    for k in res:
        assert np.array_equal(res["1hash"][0], res[k][0]) and res["1hash"][1] == res[k][1], k
    oc, oq, _, _ = orc.louvain_c(nv, src, dst, w, 100, 1e-7, 1.0)
    assert np.array_equal(res["1hash"][0], oc) and abs(res["1hash"][1] - oq) <= 1e-9


@pytest.mark.parametrize("scale", [22, 26])
def test_pagerank_rmat_golden(cg, handle, scale):
    """PageRank at the size the headline number is quoted on (RMAT-26, 1.07 G edges: bench.py's graph, built here by the library's generator)
    against the C oracle's result after 20 fixed iterations -- the fixture holds the oracle's values at 4096 vertices (the 64 of highest
    in-degree, random ones with and without in-edges: tests/golden/make_pagerank_fixture.py) and the sum: 1e-6 absolute and 2e-5 relative
    at every one of them, as the full-vector comparison at RMAT-22 (test_pagerank_config2_rmat22_vs_oracle).  RMAT-22 checks the fixture
    mechanism against that test's graph."""
    import json
    from pathlib import Path

    import torch

    f = Path(__file__).resolve().parent / "golden" / f"pagerank_rmat{scale}.json"
    if not f.exists():
        pytest.skip(f"{f.name} not committed")
    gold = json.loads(f.read_text())
    nv, ne = 1 << scale, gold["edge_factor"] << scale
    src, dst = cg.generate_rmat_edgelist(handle, scale, ne)
    g = cg.SGGraph(handle, cg.GraphProperties(is_multigraph=True), src, dst, None, store_transposed=True, renumber=True,
                   vertices_array=torch.arange(nv, dtype=torch.int32, device="cuda"))
    del src, dst
    v, pr, _ = cg.pagerank(handle, g, None, None, None, None, gold["alpha"], 0.0, gold["iterations"], False, fail_on_nonconvergence=False)
    full = torch.empty(nv, dtype=torch.float32, device="cuda")
    full[v.long()] = pr
    assert abs(float(full.double().sum()) - gold["sum"]) <= 1e-5
    got = full[torch.tensor(gold["vertices"], device="cuda")].double().cpu().numpy()
    want = np.asarray(gold["values"])
    assert np.max(np.abs(got - want)) <= 1e-6
    assert np.max(np.abs(got - want) / np.maximum(want, 1e-30)) <= 2e-5


@pytest.mark.parametrize("scale", [22, 24, 26])
def test_louvain_rmat_golden(cg, handle, scale):
    """Louvain at the sizes its timings are quoted on (RMAT-22: 65 M directed edges, the single-GPU bench line; RMAT-24; RMAT-26: 1.06 G, BASELINE
    config 5's graph): clusters (sha256 of the column: every vertex in the oracle's cluster), modularity to 1e-9 (the sum of squared cluster
    weights exceeds 2^53, so its last bits follow the summation order: the oracle adds sequentially, the library in 64 Ki chunks), against the fixtures the C oracle
    produced (tests/golden/make_louvain_fixture.py: 2.5 CPU-minutes / 10 minutes / hours of one core, hence fixtures)."""
    import hashlib
    import json
    import sys
    from pathlib import Path

    import torch

    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    from bench_louvain import undirected_rmat

    f = root / "tests" / "golden" / f"louvain_rmat{scale}.json"
    if not f.exists():
        pytest.skip(f"{f.name} not committed")
    gold = json.loads(f.read_text())
    src, dst, w = undirected_rmat(cg, handle, scale, gold["edge_factor"], seed=gold["seed"])
    assert int(src.numel()) == gold["directed_edges"]
    nv = 1 << scale
    g = cg.SGGraph(handle, cg.GraphProperties(is_symmetric=True), src, dst, w, renumber=False, vertices_array=torch.arange(nv, dtype=torch.int32, device="cuda"))
    del src, dst, w
    v, c, q = cg.louvain(handle, g, 100, 1e-7, 1.0, False)
    (c,) = by_vertex(v, c)
    assert abs(q - gold["modularity"]) <= 1e-9
    assert int(np.unique(c).size) == gold["clusters"]
    assert hashlib.sha256(np.ascontiguousarray(c, np.int32).tobytes()).hexdigest() == gold["clusters_sha256"]


def test_louvain_two_to_the_31_edges(cg, handle):
    """Round 6: Louvain on a graph of more than 2^31 directed edges (undirected simple RMAT-27, edge factor 9: 2.4 G stored edges) -- the level's
    edge positions are unsigned 32-bit words on the single-GPU path, as in graph construction and the traversals (rounds 1-5 refused such a graph;
    the reference needs 64-bit edge types for it: cpp/src/c_api/graph_sg.cpp:745-779).  The oracle cannot run at this size (hours); checked through
    size-independent properties: the returned modularity equals the modularity of the returned clustering recomputed from the edge list (torch, fp64;
    integer weights: every sum is exact), the full run is no worse than its own first level, and a second run returns the same bits."""
    import sys
    from pathlib import Path

    import torch

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    scale, nv = 27, 1 << 27
    # the construction of undirected_rmat without its final sort by (src, dst) (torch sorts hold fewer than 2^31 elements; creation orders the edges itself)
    src, dst = cg.generate_rmat_edgelist(handle, scale, 9 << scale, seed=5)
    s, d = src.to(torch.int64), dst.to(torch.int64)
    del src, dst
    keep = s != d
    key = torch.minimum(s, d)[keep] << 32 | torch.maximum(s, d)[keep]
    del s, d, keep
    key = torch.unique(key)
    lo, hi = (key >> 32).to(torch.int32), (key & 0xFFFFFFFF).to(torch.int32)
    del key
    wt = (1 + (lo.long() * 7 + hi.long() * 13) % 8).to(torch.float32)
    src, dst, w = torch.cat([lo, hi]), torch.cat([hi, lo]), torch.cat([wt, wt])
    del lo, hi, wt
    ne = int(src.numel())
    assert (1 << 31) < ne < (1 << 32) - 4097
    torch.cuda.empty_cache()
    g = cg.SGGraph(handle, cg.GraphProperties(is_symmetric=True), src, dst, w, renumber=False, vertices_array=torch.arange(nv, dtype=torch.int32, device="cuda"))
    v, c, q = cg.louvain(handle, g, 100, 1e-7, 1.0, False)
    cl = torch.empty(nv, dtype=torch.int64, device="cuda")
    cl[v.to(torch.int64)] = c.to(torch.int64)
    assert int(cl.min()) >= 0 and int(cl.max()) < nv
    m = float(w.double().sum())
    # integer sums (the weights are 1..8): torch's fp64 scatter-adds are compare-and-swap loops, and a hub or a giant cluster queues 10^8 of them on one address
    internal, k = 0, torch.zeros(nv, dtype=torch.int64, device="cuda")
    step = 1 << 28
    for lo in range(0, ne, step):  # (in slices: the whole list as int64 would take 60 GB of temporaries)
        s, d, ww = src[lo:lo + step].long(), dst[lo:lo + step].long(), w[lo:lo + step].long()
        internal += int((ww * (cl[s] == cl[d])).sum())
        k.index_add_(0, s, ww)  # vertex weights
        del s, d, ww
    a = torch.zeros(nv, dtype=torch.int64, device="cuda").index_add_(0, cl, k).double()  # cluster weights
    q_torch = internal / m - float((a * a).sum()) / (m * m)
    assert abs(q_torch - q) <= 1e-9, (q_torch, q)
    v1, c1, q1 = cg.louvain(handle, g, 1, 1e-7, 1.0, False)
    assert q >= q1 > 0.0
    v2, c2, q2 = cg.louvain(handle, g, 100, 1e-7, 1.0, False)
    assert q2 == q and torch.equal(v2, v) and torch.equal(c2, c)
    del g, src, dst, w, cl, a, k
    torch.cuda.empty_cache()


def test_capi_generators_edge_columns_and_decompress(cg, handle):
    """cugraph_generate_rmat_edgelists / _edge_ids / _edge_types (graph_generators.h), cugraph_data_type_id_from_dlpack,
    cugraph_graph_create_mg on a one-rank handle and cugraph_decompress_to_edgelist (external ids, by-source order)."""
    import ctypes as C

    import torch

    from cugraph_amd import _capi
    from cugraph_amd.pylib import _View, assert_success, copy_to_torch

    l = _capi.lib()
    hp = handle.c_resource_handle_ptr
    err, rng, lst = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert_success(l.cugraph_rng_state_create(hp, 42, C.byref(rng), C.byref(err)), err, "rng")
    assert_success(l.cugraph_generate_rmat_edgelists(hp, rng, 5, 4, 9, 16, 1, 0, 0, 0, C.byref(lst), C.byref(err)), err, "rmat_edgelists")
    assert l.cugraph_coo_list_size(lst) == 5
    for i in range(5):
        coo = C.c_void_p(l.cugraph_coo_list_element(lst, i))
        s = copy_to_torch(hp, l.cugraph_coo_get_sources(coo))
        d = copy_to_torch(hp, l.cugraph_coo_get_destinations(coo))
        scale = s.numel() // 16
        assert 4 <= scale <= 9 and s.numel() == scale * 16 == d.numel()           # the reference passes scale * edge_factor edges
        assert int(s.max()) < (1 << scale) and int(d.max()) < (1 << scale) and int(s.min()) >= 0
        assert l.cugraph_coo_get_edge_id(coo) is None
        assert_success(l.cugraph_generate_edge_ids(hp, coo, 0, C.byref(err)), err, "edge_ids")
        assert_success(l.cugraph_generate_edge_types(hp, rng, coo, 3, 7, C.byref(err)), err, "edge_types")
        ids = copy_to_torch(hp, l.cugraph_coo_get_edge_id(coo))
        ty = copy_to_torch(hp, l.cugraph_coo_get_edge_type(coo))
        assert torch.equal(ids.cpu(), torch.arange(s.numel(), dtype=torch.int32)) and int(ty.min()) >= 3 and int(ty.max()) <= 7
    l.cugraph_coo_list_free(lst)
    l.cugraph_rng_state_free(rng)

    class DL(C.Structure):
        _fields_ = [("code", C.c_uint8), ("bits", C.c_uint8), ("lanes", C.c_uint16)]
    out = C.c_int(-1)
    for code, bits, want in ((0, 32, _capi.INT32), (0, 64, _capi.INT64), (2, 32, _capi.FLOAT32), (2, 64, _capi.FLOAT64), (1, 8, _capi.UINT8), (6, 8, _capi.BOOL)):
        assert l.cugraph_data_type_id_from_dlpack(C.cast(C.byref(DL(code, bits, 1)), C.c_void_p), C.byref(out), C.byref(err)) == 0 and out.value == want
    assert l.cugraph_data_type_id_from_dlpack(C.cast(C.byref(DL(0, 32, 4)), C.c_void_p), C.byref(out), C.byref(err)) == _capi.CUGRAPH_UNSUPPORTED_TYPE_COMBINATION
    l.cugraph_error_free(err)
    assert l.cugraph_data_type_id_from_dlpack(C.cast(C.byref(DL(2, 16, 1)), C.c_void_p), C.byref(out), C.byref(err)) == _capi.CUGRAPH_INVALID_INPUT
    l.cugraph_error_free(err)

    # create_mg with the edge list split over two arrays == create_sg on the whole list; decompress returns the input multiset
    src = np.array([0, 1, 1, 2, 2, 2, 3, 4], np.int32) * 10 + 5
    dst = np.array([1, 3, 4, 0, 1, 3, 5, 5], np.int32) * 10 + 5
    w = np.array([0.1, 2.1, 1.1, 5.1, 3.1, 4.1, 7.2, 3.2], np.float32)
    ts, td, tw = T(src), T(dst), T(w)
    views = [[_View(t[:3]), _View(t[3:])] for t in (ts, td, tw)]
    arr = lambda vs: (C.c_void_p * 2)(*[v.ptr for v in vs])
    props = _capi.GraphPropertiesStruct(0, 0)
    g = C.c_void_p()
    code = l.cugraph_graph_create_mg(hp, C.byref(props), None, arr(views[0]), arr(views[1]), arr(views[2]), None, None, 1, 2, 0, 0, 0, 0, C.byref(g), C.byref(err))
    assert_success(code, err, "cugraph_graph_create_mg")
    el = C.c_void_p()
    assert_success(l.cugraph_decompress_to_edgelist(hp, g, 0, C.byref(el), C.byref(err)), err, "decompress")
    es = copy_to_torch(hp, l.cugraph_edgelist_get_sources(el)).cpu().numpy()
    ed = copy_to_torch(hp, l.cugraph_edgelist_get_destinations(el)).cpu().numpy()
    ew = copy_to_torch(hp, l.cugraph_edgelist_get_edge_weights(el)).cpu().numpy()
    assert l.cugraph_edgelist_get_edge_ids(el) is None and l.cugraph_edgelist_get_edge_offsets(el) is None
    assert sorted(zip(es.tolist(), ed.tolist(), ew.tolist())) == sorted(zip(src.tolist(), dst.tolist(), w.tolist()))
    l.cugraph_edgelist_free(el)
    l.cugraph_graph_free(g)
    for vs in views:
        for v in vs:
            v.free()


@pytest.mark.parametrize("transposed,id_dtype", [(False, np.int32), (True, np.int64)])
def test_edge_ids_and_types_travel_with_their_edges(cg, handle, orc, transposed, id_dtype):
    """Edge ids / edge type ids given at graph creation are stored with their edges and come back from cugraph_decompress_to_edgelist
    (graph_sg.cpp:781-830; c_api/decompress_to_edgelist.cpp): the id returned next to an edge names exactly that input edge -- also when the
    by-source storage is derived from a transposed graph (a second sort) and after renumbering.  With a flag that rewrites the edge list the
    combination is refused, not silently dropped."""
    import ctypes as C

    from cugraph_amd import _capi
    from cugraph_amd.pylib import assert_success, copy_to_torch

    s, d = rmat_graph(orc, 10)
    ne = s.size
    w = np.random.default_rng(3).random(ne).astype(np.float32)
    ids = (np.random.default_rng(4).permutation(ne) + 1000).astype(id_dtype)   # arbitrary distinct ids
    types = (np.arange(ne) % 5).astype(np.int32)
    g = cg.SGGraph(handle, cg.GraphProperties(is_multigraph=True), T(s), T(d), T(w), store_transposed=transposed, renumber=True, edge_id_array=T(ids),
                   edge_type_array=T(types))
    if transposed:  # exercise the derived orientation once more through an algorithm that flips it
        cg.pagerank(handle, g, None, None, None, None, 0.85, 0.0, 2, False, fail_on_nonconvergence=False)
    l, hp = _capi.lib(), handle.c_resource_handle_ptr
    el, err = C.c_void_p(), C.c_void_p()
    assert_success(l.cugraph_decompress_to_edgelist(hp, g.c_graph_ptr, 0, C.byref(el), C.byref(err)), err, "decompress")
    es = copy_to_torch(hp, l.cugraph_edgelist_get_sources(el)).cpu().numpy()
    ed = copy_to_torch(hp, l.cugraph_edgelist_get_destinations(el)).cpu().numpy()
    ew = copy_to_torch(hp, l.cugraph_edgelist_get_edge_weights(el)).cpu().numpy()
    ei = copy_to_torch(hp, l.cugraph_edgelist_get_edge_ids(el)).cpu().numpy()
    et = copy_to_torch(hp, l.cugraph_edgelist_get_edge_type_ids(el)).cpu().numpy()
    l.cugraph_edgelist_free(el)
    assert ei.dtype == id_dtype and sorted(ei.tolist()) == sorted(ids.tolist())
    where = np.empty(ne + 1000, np.int64)
    where[ids] = np.arange(ne)
    k = where[ei]   # the input edge every returned row claims to be
    assert np.array_equal(es, s[k]) and np.array_equal(ed, d[k]) and np.array_equal(ew, w[k]) and np.array_equal(et, types[k])
    with pytest.raises(Exception, match="NOT_IMPLEMENTED|not supported"):
        cg.SGGraph(handle, cg.GraphProperties(), T(s), T(d), T(w), renumber=True, edge_id_array=T(ids), drop_self_loops=True)


# ----------------------------------------------------------------------- INT64 ids, sparse ids (outer_ids.hip)
def _ext(ids, mapping, dtype):
    return T(np.asarray([mapping[i] for i in ids]), dtype)


@pytest.mark.parametrize("kind", ["int64_dense", "int64_sparse", "int32_sparse", "int64_norenumber"])
def test_goldens_with_int64_and_sparse_ids(cg, handle, golden, kind):
    """The reference's C-API goldens (pagerank_test.c, bfs_test.c, sssp_test.c) with the vertex ids as INT64 columns
    (graph_sg.cpp:745-779 instantiates int64 graphs; python-cugraph's default column type) and with ids spread over ranges far
    too wide for a dense table (renumber_utils_impl.cuh:333-660 uses a hash map): every vertex-id column comes back in the
    caller's type and id space, BFS distances in the vertex type; has_vertex / degrees / decompress_to_edgelist follow."""
    import ctypes as C

    import torch

    from cugraph_amd import _capi
    from cugraph_amd.pylib import assert_success, copy_to_torch

    dtype = np.int32 if kind == "int32_sparse" else np.int64
    if kind == "int32_sparse":
        m = {0: 7, 1: 10**9, 2: 2 * 10**9, 3: 123456789, 4: 42, 5: 1999999999}
    elif kind == "int64_sparse":
        m = {0: 7, 1: 10**9, 2: 2 * 10**9, 3: 3 * 10**12, 4: -5, 5: 2**62}
    else:
        m = {i: i for i in range(8)}
    inv = {v: k for k, v in m.items()}
    renumber = kind != "int64_norenumber"
    tdtype = torch.int32 if dtype == np.int32 else torch.int64

    def back(t):
        return np.array([inv.get(int(x), int(x)) for x in t.cpu().numpy().tolist()])

    props = cg.GraphProperties(is_multigraph=True)

    def graph(gr, transposed, wdtype=np.float32):
        return cg.SGGraph(handle, props, _ext(gr["src"], m, dtype), _ext(gr["dst"], m, dtype), T(gr["wgt"], wdtype), store_transposed=transposed, renumber=renumber)

    case = golden["c_api"]["pagerank"][0]
    g = graph(case["graph"], case["store_transposed"])
    v, pr, conv = cg.pagerank(handle, g, None, None, None, None, case["alpha"], case["epsilon"], case["max_iterations"], False, fail_on_nonconvergence=False)
    assert v.dtype == tdtype
    got = np.empty(len(case["result"]))
    got[back(v)] = pr.cpu().numpy()
    for a, b in zip(got, case["result"]):
        assert nearly_equal(float(a), b, golden["c_api"]["tolerance"])
    pc = golden["c_api"]["personalized_pagerank"][0]
    gp = graph(pc["graph"], pc["store_transposed"])
    v, pr, _ = cg.personalized_pagerank(handle, gp, None, None, None, None, _ext(pc["pers_vertices"], m, dtype), T(pc["pers_values"], np.float32),
                                        pc["alpha"], pc["epsilon"], pc["max_iterations"], False, fail_on_nonconvergence=False)
    got = np.empty(len(pc["result"]))
    got[back(v)] = pr.cpu().numpy()
    for a, b in zip(got, pc["result"]):
        assert nearly_equal(float(a), b, golden["c_api"]["tolerance"])
    bc = golden["c_api"]["bfs"][0]
    gb = graph(bc["graph"], bc["store_transposed"])
    d, p, v = cg.bfs(handle, gb, _ext(bc["seeds"], m, dtype), False, bc["depth_limit"], True, False)
    assert d.dtype == tdtype and v.dtype == tdtype and p.dtype == tdtype
    order = np.argsort(back(v))
    tmax = int(np.iinfo(dtype).max)
    assert d.cpu().numpy()[order].tolist() == [x if x != 2147483647 else tmax for x in bc["distances"]]
    pb, db = back(p)[order].tolist(), bc["distances"]
    if "sparse" in kind:  # the tie-break among valid parents follows the id order, which the sparse mapping permutes: any valid parent
        edges = set(zip(bc["graph"]["src"], bc["graph"]["dst"]))
        for x, par in enumerate(pb):
            assert (par == -1 and (db[x] == 0 or db[x] == 2147483647)) or ((par, x) in edges and db[par] + 1 == db[x])
    else:
        assert pb == bc["predecessors"]
    with pytest.raises(ValueError):  # a source that is not a vertex
        cg.bfs(handle, gb, T([999], dtype), False, 0, True, False)
    sc = golden["c_api"]["sssp"][0]
    gs = graph(sc["graph"], sc["store_transposed"], np.dtype(sc["dtype"]))
    v, d, p = cg.sssp(handle, gs, m[sc["source"]], float(np.finfo(np.dtype(sc["dtype"])).max), True, False)
    order = np.argsort(back(v))
    for a, b in zip(d.cpu().numpy()[order], sc["distances"]):
        assert nearly_equal(float(a), b, golden["c_api"]["tolerance"])
    ps, ds = back(p)[order].tolist(), d.cpu().numpy()[order].tolist()
    if "sparse" in kind:
        wmap = {(a, b): w for a, b, w in zip(sc["graph"]["src"], sc["graph"]["dst"], sc["graph"]["wgt"])}
        for x, par in enumerate(ps):
            assert (par == -1) or ((par, x) in wmap and nearly_equal(ds[par] + wmap[(par, x)], ds[x], 1e-5))
    else:
        assert ps == sc["predecessors"]
    hv = cg.has_vertex(handle, gb, T([m[0], m[5], 31337], dtype))
    assert hv.cpu().numpy().tolist() == [True, True, False]
    l, hp, err = _capi.lib(), handle.c_resource_handle_ptr, C.c_void_p()
    el = C.c_void_p()
    assert_success(l.cugraph_decompress_to_edgelist(hp, gb.c_graph_ptr, 0, C.byref(el), C.byref(err)), err, "decompress")
    es = copy_to_torch(hp, l.cugraph_edgelist_get_sources(el))
    ed = copy_to_torch(hp, l.cugraph_edgelist_get_destinations(el))
    assert es.dtype == tdtype
    assert sorted(zip(back(es).tolist(), back(ed).tolist())) == sorted(zip(bc["graph"]["src"], bc["graph"]["dst"]))
    l.cugraph_edgelist_free(el)
    res = C.c_void_p()
    assert_success(l.cugraph_degrees(hp, gb.c_graph_ptr, None, 0, C.byref(res), C.byref(err)), err, "degrees")
    dv = copy_to_torch(hp, l.cugraph_degrees_result_get_vertices(res))
    do = copy_to_torch(hp, l.cugraph_degrees_result_get_out_degrees(res))
    outd = np.zeros(6, np.int64)
    outd[back(dv)] = do.cpu().numpy()
    assert outd.tolist() == np.bincount(bc["graph"]["src"], minlength=6).tolist()
    l.cugraph_degrees_result_free(res)


def test_pagerank_argument_checks(cg, handle):
    """pagerank_impl.cuh:78-117: alpha in [0, 1], epsilon >= 0, a personalization vector must not be empty; with
    do_expensive_check negative edge weights / out-weight sums are rejected."""
    g = make_graph(cg, handle, [0, 1, 2], [1, 2, 0], [1.0, -2.0, 1.0], transposed=True)
    with pytest.raises(ValueError, match="alpha should be in"):
        cg.pagerank(handle, g, None, None, None, None, 1.5, 1e-6, 10, False)
    with pytest.raises(ValueError, match="epsilon should be non-negative"):
        cg.pagerank(handle, g, None, None, None, None, 0.85, -1.0, 10, False)
    with pytest.raises(ValueError, match="personalization vector size should not be 0"):
        cg.personalized_pagerank(handle, g, None, None, None, None, T([], np.int32), T([], np.float32), 0.85, 1e-6, 10, False)
    with pytest.raises(ValueError, match="edge weights should have non-negative values"):
        cg.pagerank(handle, g, None, None, None, None, 0.85, 1e-6, 10, True)
    g2 = make_graph(cg, handle, [0, 1, 2], [1, 2, 0], [1.0, 2.0, 1.0], transposed=True)
    with pytest.raises(ValueError, match="outgoing edge weight sum values should be non-negative"):
        cg.pagerank(handle, g2, T([0, 1, 2], np.int32), T([1.0, -1.0, 1.0], np.float32), None, None, 0.85, 1e-6, 10, True)
    cg.pagerank(handle, g2, None, None, None, None, 0.85, 1e-6, 100, True)  # clean input passes the expensive checks


def _decompress(cg, handle, g):
    import ctypes as C

    from cugraph_amd import _capi
    from cugraph_amd.pylib import assert_success, copy_to_torch

    l, hp, err, el = _capi.lib(), handle.c_resource_handle_ptr, C.c_void_p(), C.c_void_p()
    assert_success(l.cugraph_decompress_to_edgelist(hp, g.c_graph_ptr, 0, C.byref(el), C.byref(err)), err, "decompress")
    s = copy_to_torch(hp, l.cugraph_edgelist_get_sources(el)).cpu().numpy()
    d = copy_to_torch(hp, l.cugraph_edgelist_get_destinations(el)).cpu().numpy()
    wv = l.cugraph_edgelist_get_edge_weights(el)
    w = copy_to_torch(hp, wv).cpu().numpy() if wv else None
    l.cugraph_edgelist_free(el)
    return s, d, w


def test_graph_creation_flags_vs_networkx(cg, handle):
    """Second opinion for the creation flags (graph_sg.cpp:185-248), independent of this repo's numpy restatement: the edge list
    that comes back from cugraph_decompress_to_edgelist must be what NetworkX makes of the same input --
    drop_self_loops: the input minus its loops; drop_multi_edges: one edge per (src, dst) with the MINIMUM weight
    (remove_multi_edges keep_min_value_edge, as the C API passes it); symmetrize (simple input): the edge set of
    nx.Graph(G) in both directions, weights of one-directional edges kept, reciprocal pairs averaged."""
    nx = pytest.importorskip("networkx")
    rng = np.random.default_rng(11)
    nv, ne = 200, 3000
    s = rng.integers(0, nv, ne).astype(np.int32)
    d = rng.integers(0, nv, ne).astype(np.int32)
    d[:100] = s[:100]
    w = rng.integers(1, 50, ne).astype(np.float32)
    verts = T(np.arange(nv), np.int32)
    # drop_self_loops
    g = cg.SGGraph(handle, cg.GraphProperties(is_multigraph=True), T(s), T(d), T(w), renumber=True, vertices_array=verts, drop_self_loops=True)
    gs, gd, gw = _decompress(cg, handle, g)
    keep = s != d
    assert sorted(zip(gs.tolist(), gd.tolist(), gw.tolist())) == sorted(zip(s[keep].tolist(), d[keep].tolist(), w[keep].tolist()))
    # drop_multi_edges: NetworkX DiGraph keeps one edge per pair; feed it the minimum weight per pair
    g = cg.SGGraph(handle, cg.GraphProperties(is_multigraph=False), T(s), T(d), T(w), renumber=True, vertices_array=verts, drop_multi_edges=True)
    gs, gd, gw = _decompress(cg, handle, g)
    G = nx.DiGraph()
    for a, b, c in zip(s.tolist(), d.tolist(), w.tolist()):
        if not G.has_edge(a, b) or G[a][b]["weight"] > c:
            G.add_edge(a, b, weight=c)
    assert sorted(zip(gs.tolist(), gd.tolist(), gw.tolist())) == sorted((a, b, dd["weight"]) for a, b, dd in G.edges(data=True))
    # symmetrize on the simple graph (no multi-edges): undirected closure
    es = np.array([e[0] for e in G.edges()], np.int32)
    ed = np.array([e[1] for e in G.edges()], np.int32)
    ew = np.array([G[a][b]["weight"] for a, b in G.edges()], np.float32)
    g = cg.SGGraph(handle, cg.GraphProperties(is_symmetric=True, is_multigraph=False), T(es), T(ed), T(ew), renumber=True, vertices_array=verts,
                   symmetrize=True)
    gs, gd, gw = _decompress(cg, handle, g)
    U = nx.Graph(G)  # undirected closure; self-loops stay single
    want = {}
    for a, b in U.edges():
        fw = G[a][b]["weight"] if G.has_edge(a, b) else None
        bw = G[b][a]["weight"] if G.has_edge(b, a) else None
        wt = (fw + bw) / 2 if (fw is not None and bw is not None and a != b) else (fw if fw is not None else bw)
        want[(a, b)] = wt
        want[(b, a)] = wt
    got = {(a, b): c for a, b, c in zip(gs.tolist(), gd.tolist(), gw.tolist())}
    assert len(got) == len(gs), "symmetrize must not create duplicate edges on a simple graph"
    assert set(got) == set(want)
    for k, v in want.items():
        assert abs(got[k] - v) <= 1e-6 * max(1.0, abs(v)), (k, got[k], v)
    # and the result is symmetric in NetworkX's eyes too
    H = nx.DiGraph(); H.add_edges_from(zip(gs.tolist(), gd.tolist()))
    assert all(H.has_edge(b, a) for a, b in H.edges())


def test_device_array_release_and_pool_trim(cg, handle):
    """cugraph_type_erased_device_array_release (cpp/include/cugraph_c/array.h:85): the caller takes the device block over and
    frees it with hipFree; the array object stays a valid empty array.  Also: the cache of freed device blocks stays below its cap
    and cugraph_amd_memory_pool_trim_large / _trim hand it back."""
    import ctypes as C

    import torch

    l, capi = cg.pylib.capi.lib(), cg.pylib.capi
    arr, err = C.c_void_p(), C.c_void_p()
    assert l.cugraph_type_erased_device_array_create(handle.c_resource_handle_ptr, 1000, capi.INT32, C.byref(arr), C.byref(err)) == 0
    view = l.cugraph_type_erased_device_array_view(arr)
    host = np.arange(1000, dtype=np.int32)
    assert l.cugraph_type_erased_device_array_view_copy_from_host(handle.c_resource_handle_ptr, view, host.ctypes.data_as(C.c_void_p), C.byref(err)) == 0
    ptr = l.cugraph_type_erased_device_array_view_pointer(view)
    l.cugraph_type_erased_device_array_view_free(view)
    raw = l.cugraph_type_erased_device_array_release(arr)
    assert raw == ptr
    v2 = l.cugraph_type_erased_device_array_view(arr)
    assert l.cugraph_type_erased_device_array_view_size(v2) == 0
    l.cugraph_type_erased_device_array_view_free(v2)
    l.cugraph_type_erased_device_array_free(arr)  # must not free `raw`
    back = np.empty(1000, np.int32)
    borrowed = l.cugraph_type_erased_device_array_view_create(raw, 1000, capi.INT32)
    assert l.cugraph_type_erased_device_array_view_copy_to_host(handle.c_resource_handle_ptr, back.ctypes.data_as(C.c_void_p), borrowed, C.byref(err)) == 0
    l.cugraph_type_erased_device_array_view_free(borrowed)
    assert np.array_equal(back, host)
    hip = C.CDLL("libamdhip64.so")
    hip.hipFree.argtypes = [C.c_void_p]
    assert hip.hipFree(raw) == 0
    # large temporaries of a graph build leave the cache when the build returns
    src, dst = cg.generate_rmat_edgelist(handle, 22, 16 << 22)  # 512 MiB of edge-list temporaries and more
    g = cg.SGGraph(handle, cg.GraphProperties(is_multigraph=True), src, dst, None, store_transposed=True, renumber=True)
    del src, dst
    cached = l.cugraph_amd_memory_pool_cached_bytes()
    assert cached <= (128 << 30)
    freed_large = l.cugraph_amd_memory_pool_trim_large(256 << 20)  # the sort buffers of the build (512 MiB of keys and more)
    assert freed_large > 0 and l.cugraph_amd_memory_pool_cached_bytes() == cached - freed_large
    n_before = l.cugraph_amd_memory_pool_trim()
    assert n_before == cached - freed_large and l.cugraph_amd_memory_pool_cached_bytes() == 0
    v, pr, _ = cg.pagerank(handle, g, None, None, None, None, 0.85, 0.0, 3, False, fail_on_nonconvergence=False)
    assert abs(float(pr.sum()) - 1.0) < 1e-4
    torch.cuda.synchronize()


@pytest.mark.parametrize("chunk,static,tile,weighted", [
    (1, False, 256, False), (3, False, 256, False), (7, True, 256, False), (4, True, 1024, True), (16, True, 0, False), (2, False, 4096, True)])
def test_pagerank_phase1_schedules_agree(cg, handle, orc, monkeypatch, chunk, static, tile, weighted):
    """Phase 1's schedule (round 3): pre-assigned ("static") chunk ranges per workgroup in front of the dynamically drawn chunks.
    Forced here at a size the oracle checks: chunks of several items, a hub row (one run over many wavefront ranges), runs that
    end exactly at range ends, tiles whose last item is mostly padding.  Every schedule must give the oracle's vector."""
    monkeypatch.setenv("CUGRAPH_AMD_PAGERANK_KERNEL", "tiled")
    monkeypatch.setenv("CUGRAPH_AMD_TILED_REBUILD", "1")
    monkeypatch.setenv("CUGRAPH_AMD_TP_CHUNK", str(chunk))
    if static:
        monkeypatch.setenv("CUGRAPH_AMD_TP_STATIC_FORCE", "1")
        monkeypatch.setenv("CUGRAPH_AMD_TP_STATIC_FRAC", "0.7")
    else:
        monkeypatch.setenv("CUGRAPH_AMD_TP_STATIC_FRAC", "0")
    prev = handle.set_pagerank_hot_tile(tile if tile else -1)
    rng = np.random.default_rng(5)
    nv = 20000
    hub_in = rng.integers(0, 700, 200003)                       # 200003 in-edges of vertex 3 from 700 sources: long runs in few tiles
    exact = np.repeat(np.arange(1024), 16)                      # 1024 rows x 16 in-edges from sources 0..15: runs that end at range ends
    s = np.concatenate([hub_in, rng.integers(0, nv, 300000), np.tile(np.arange(16), 1024), np.arange(100, 1100)])
    d = np.concatenate([np.full(hub_in.size, 3), rng.integers(0, nv, 300000), 5000 + exact, np.arange(2000, 3000)])
    w = int_weights(s.size, seed=8) if weighted else None
    g = make_graph(cg, handle, s, d, w, transposed=True, renumber=True, vertices=np.arange(nv))
    try:
        v, pr, _ = cg.pagerank(handle, g, None, None, None, None, 0.85, 0.0, 12, False, fail_on_nonconvergence=False)
    finally:
        handle.set_pagerank_hot_tile(prev)
    off, idx, ww = orc.coo_to_cs(nv, d.astype(np.int32), s.astype(np.int32), None if w is None else w.astype(np.float32))
    truth, _, _ = orc.pagerank(nv, off, idx, ww, 0.85, 0.0, 12, acc64=True)
    got = by_vertex(v, pr)[0]
    assert np.max(np.abs(got - truth) / truth) <= 2e-5
    assert abs(float(got.sum()) - 1.0) <= 1e-5


def test_pagerank_two_to_the_31_edges(cg, handle):
    """Round 5: RMAT-27 (2^31 directed edges, 134 M vertices) on one GPU through the column-tiled plan -- its edge arrays are addressed from
    64-bit per-wavefront bases, so only edge POSITIONS have to fit 32 bits (the reference needs 64-bit vertex / edge types for such a graph:
    cpp/src/c_api/graph_sg.cpp:745-779).  The oracle cannot run at this size; the vector is checked the way bench.py checks the timed one:
    mass, and one more library iteration against one explicit fp64 power iteration in torch on the regenerated edge list."""
    import sys
    from pathlib import Path

    import torch

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench

    scale = 27
    nv, ne = 1 << scale, 16 << scale
    src, dst = cg.generate_rmat_edgelist(handle, scale, ne)
    g = cg.SGGraph(handle, cg.GraphProperties(is_multigraph=True), src, dst, None, store_transposed=True, renumber=True,
                   vertices_array=torch.arange(nv, dtype=torch.int32, device="cuda"))
    del src, dst
    plan = cg.PageRankPlan(handle, g, 0.85)
    plan.step(6)
    chk = bench.check_result(cg, handle, plan, scale, ne, nv)
    assert chk["ok"], chk
    del plan, g
    torch.cuda.empty_cache()
