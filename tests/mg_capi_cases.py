"""What a rank of tests/test_mg_capi.py runs through the reference's own entry points on a communicator handle
(cugraph_graph_create_mg -> cugraph_pagerank / cugraph_bfs / cugraph_sssp / cugraph_louvain).  Every rank builds ITS slice of the same
seeded RMAT edge list, saves (vertices, values) of the vertices it gets back; the test assembles them and compares with the oracle."""
import os

import numpy as np
import torch

# CUGRAPH_AMD_TEST_WIDE_IDS=1: every vertex id crosses the C API as INT64, spread out and shifted past 2^40 (v -> v * WIDE_MUL + WIDE_OFF, monotone:
# minimum-external-id tie-breaks are unchanged); what comes back is mapped home before it is saved, so the checks of the int32 cases apply as they are
WIDE = os.environ.get("CUGRAPH_AMD_TEST_WIDE_IDS") == "1"
WIDE_MUL, WIDE_OFF = 1000003, 1 << 40


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).cuda()


def wide_of(a):
    return np.asarray(a).astype(np.int64) * WIDE_MUL + WIDE_OFF


def X(a):
    """a column of vertex ids on its way into the library"""
    return T(wide_of(a)) if WIDE else T(a)


def xid(v):
    return int(v) * WIDE_MUL + WIDE_OFF if WIDE else int(v)


def U(t):
    """a column of vertex ids that came out of the library (negative markers -- no predecessor, path padding -- stay as they are), as int32"""
    a = t.cpu().numpy()
    if not WIDE:
        return a
    assert a.dtype == np.int64, a.dtype
    ok = a >= 0
    assert ((a[ok] - WIDE_OFF) % WIDE_MUL == 0).all(), "an id that is not one of the graph's came back"
    return np.where(ok, (a - WIDE_OFF) // WIDE_MUL, a).astype(np.int32)


def UD(t):
    """BFS hop counts (typed like the vertices: INT64 with INT64_MAX = unreached in the wide case) as the int32 column the checks expect"""
    a = t.cpu().numpy()
    if not WIDE:
        return a
    assert a.dtype == np.int64, a.dtype
    return np.where(a == np.iinfo(np.int64).max, np.iinfo(np.int32).max, a).astype(np.int32)


def rmat_slice(scale, rank, size, edge_factor=16, seed=0):
    from oracle import oracle as orc  # the checker's generator = the library's (smoke() compares them); input only

    ne = edge_factor << scale
    per = (ne + size - 1) // size
    first = min(rank * per, ne)
    return orc.rmat(scale, max(0, min(per, ne - first)), seed=seed, first_edge=first), first


def ppr_inputs(scale):
    """personalization pairs (with a repeated vertex), an initial guess and precomputed out-weight sums (= the true out-degrees: the result
    must equal the run without them) for the RMAT graph of `scale`; the same on every rank and in the checker"""
    nv = 1 << scale
    rng = np.random.default_rng(11)
    pv = rng.choice(nv, size=97, replace=False).astype(np.int32)
    pv = np.concatenate([pv, pv[:3]])
    pval = rng.random(pv.size).astype(np.float32) + np.float32(0.1)
    guess = (rng.random(nv).astype(np.float32) + np.float32(0.5)) / np.float32(nv)
    (s, _), _ = rmat_slice(scale, 0, 1)
    outw = np.bincount(s, minlength=nv).astype(np.float32)
    return pv, pval, guess, outw


def run(what, cg, h, comm, rank, size, outdir, args):
    out = {}
    if os.environ.get("CUGRAPH_AMD_TEST_HOT_TILE"):  # small source tiles: a rank's gather window then spans many of them (the two-chunk exchange splits phase 1 by source tile)
        h.set_pagerank_hot_tile(int(os.environ["CUGRAPH_AMD_TEST_HOT_TILE"]))
    if what == "pagerank":
        scale, max_iter, eps, weighted = int(args[0]), int(args[1]), float(args[2]), args[3] == "w"
        (s, d), first = rmat_slice(scale, rank, size)
        w = None
        if weighted:
            w = np.random.default_rng(1).integers(1, 9, size=16 << scale).astype(np.float32)[first: first + s.size].copy()
        # isolated ids are vertices only when somebody lists them (graph_mg.cpp:326: vertices are optional): every rank lists a slice
        verts = np.arange(rank, 1 << scale, size, dtype=np.int32)
        g = cg.MGGraph(h, cg.GraphProperties(is_multigraph=True), [X(s)], [X(d)], None if w is None else [T(w)], store_transposed=True, vertices_array=[X(verts)])
        v, x, conv = cg.pagerank(h, g, None, None, None, None, 0.85, eps, max_iter, False, fail_on_nonconvergence=False)
        np.savez(outdir / f"rank{rank}.npz", v=U(v), x=x.cpu().numpy())
        out["rows"] = int(v.numel())
        out["converged"] = bool(conv) if conv is not None else None
        # a second call on the same graph reuses the partition and must give the same answer
        v2, x2, _ = cg.pagerank(h, g, None, None, None, None, 0.85, eps, max_iter, False, fail_on_nonconvergence=False)
        out["repeat_equal"] = bool(torch.equal(v, v2) and torch.equal(x, x2))
        if len(args) > 4 and int(args[4]) > 0:
            # many calls on one communicator: every call builds and frees a plan; the plan's signal channel goes back to the communicator
            # (64 channels per communicator -- a loop of PageRanks / traversals used to fail after ~60 calls)
            ok = True
            for i in range(int(args[4])):
                v3, x3, _ = cg.pagerank(h, g, None, None, None, None, 0.85, eps, 2, False, fail_on_nonconvergence=False)
                ok = ok and v3.numel() == v.numel()
            v3, x3, _ = cg.pagerank(h, g, None, None, None, None, 0.85, eps, max_iter, False, fail_on_nonconvergence=False)
            out["many_calls_equal"] = bool(ok and torch.equal(x, x3))
        del g
    elif what == "ppr":
        # the optional arguments of cugraph_personalized_pagerank on a multi-GPU graph: every rank hands over a SLICE of each (vertices, values)
        # list, naming vertices other ranks own (the library routes the pairs)
        scale, max_iter = int(args[0]), int(args[1])
        (s, d), first = rmat_slice(scale, rank, size)
        nv = 1 << scale
        verts = np.arange(rank, nv, size, dtype=np.int32)
        g = cg.MGGraph(h, cg.GraphProperties(is_multigraph=True), [X(s)], [X(d)], None, store_transposed=True, vertices_array=[X(verts)])
        pv, pval, guess, outw = ppr_inputs(scale)
        cut = lambda a: a[(rank * a.size) // size: ((rank + 1) * a.size) // size].copy()  # noqa: E731
        allv = np.arange(nv, dtype=np.int32)[::-1].copy()  # (descending: a rank's slice names mostly other ranks' vertices)
        v, x, _ = cg.personalized_pagerank(h, g, X(cut(allv)), T(cut(outw[allv])), X(cut(allv)), T(cut(guess[allv])), X(cut(pv)), T(cut(pval)), 0.85, 0.0, max_iter,
                                           False, fail_on_nonconvergence=False)
        np.savez(outdir / f"rank{rank}.npz", v=U(v), x=x.cpu().numpy())
        out["rows"] = int(v.numel())
        # a vertex that is not in the graph: INVALID_INPUT on every rank
        try:
            cg.personalized_pagerank(h, g, None, None, None, None, X(np.array([nv + 5], np.int32) if rank == 0 else np.zeros(0, np.int32)),
                                     T(np.array([1.0], np.float32) if rank == 0 else np.zeros(0, np.float32)), 0.85, 0.0, 2, False, fail_on_nonconvergence=False)
            out["bad_vertex"] = "accepted"
        except Exception as e:  # noqa: BLE001
            out["bad_vertex"] = str(e)
        del g
    elif what == "agree":
        # round 6: a bad or different argument on ONE rank of a collective call: every rank fails, at once, and the communicator stays usable
        import time

        scale = int(args[0])
        (s, d), first = rmat_slice(scale, rank, size)
        verts = np.arange(rank, 1 << scale, size, dtype=np.int32)
        w = np.ones(s.size, np.float32)
        g = cg.MGGraph(h, cg.GraphProperties(is_multigraph=True), [X(s)], [X(d)], [T(w)], store_transposed=False, vertices_array=[X(verts)])
        msgs, t0 = {}, time.perf_counter()
        for name, call in (("alpha", lambda: cg.pagerank(h, g, None, None, None, None, 2.0 if rank == 0 else 0.85, 0.0, 3, False, fail_on_nonconvergence=False)),
                           ("epsilon", lambda: cg.pagerank(h, g, None, None, None, None, 0.85, -1.0 if rank == size - 1 else 0.0, 3, False, fail_on_nonconvergence=False)),
                           ("iterations", lambda: cg.pagerank(h, g, None, None, None, None, 0.85, 0.0, 3 + (rank == 0), False, fail_on_nonconvergence=False)),
                           ("source", lambda: cg.sssp(h, g, xid(int(s[0]) if rank == 0 else int(d[0])), 3.0e38, False, False)),
                           ("depth", lambda: cg.bfs(h, g, X(np.array([int(s[0])], np.int32) if rank == 0 else np.zeros(0, np.int32)), False, 2 + rank, False, False)),
                           # a vertex column of the wrong id type on ONE rank (the degree calls and extract_paths are collective on a multi-GPU graph too)
                           ("degrees", lambda: cg.out_degrees(h, g, T(np.zeros(1, np.int32 if WIDE else np.int64)) if rank == 0 else X(verts[:1]))),
                           ("paths", lambda: cg.bfs_extract_paths(h, g, X(np.array([int(s[0])], np.int32) if rank == 0 else np.zeros(0, np.int32)),
                                                                  T(np.zeros(1, np.int32 if WIDE else np.int64)) if rank == size - 1 else X(verts[:1])))):
            try:
                call()
                msgs[name] = "accepted"
            except Exception as e:  # noqa: BLE001
                msgs[name] = str(e)
        out["seconds"] = time.perf_counter() - t0
        out["messages"] = msgs
        v, x, _ = cg.pagerank(h, g, None, None, None, None, 0.85, 0.0, 3, False, fail_on_nonconvergence=False)  # the session still works
        out["rows_after"] = int(v.numel())
        del g
    elif what in ("bfs", "sssp"):
        scale, n_roots, with_pred = int(args[0]), int(args[1]), args[2] == "p"
        (s, d), first = rmat_slice(scale, rank, size)
        nv = 1 << scale
        verts = np.arange(rank, nv, size, dtype=np.int32)
        w = None
        if what == "sssp":
            kind = args[3]
            wall = (np.ones(16 << scale, np.float32) if kind == "unit" else np.random.default_rng(1).integers(1, 256, size=16 << scale).astype(np.float32))
            if kind == "f64":  # FLOAT64 weights (the plain exchange loop): the integer weights + a fraction that float32 cannot hold
                wall = wall.astype(np.float64) + 1.0 / 3.0
            w = wall[first: first + s.size].copy()
        g = cg.MGGraph(h, cg.GraphProperties(is_multigraph=True), [X(s)], [X(d)], None if w is None else [T(w)], store_transposed=False, vertices_array=[X(verts)])
        # roots: the same list on every rank (vertices with out-edges, fixed seed); BFS hands every rank a SLICE of it (the union is the source set)
        all_s, _ = rmat_slice(scale, 0, 1)
        cand = np.unique(all_s[0])
        roots = np.random.default_rng(7).choice(cand, size=n_roots, replace=False).astype(np.int32)
        res = {}
        if what == "bfs":
            depth = int(args[3]) if len(args) > 3 else 0
            for k, root in enumerate(roots):
                mine = np.array([root], np.int32) if k % size == rank else np.zeros(0, np.int32)  # one rank names the source
                dist, pred, v = cg.bfs(h, g, X(mine), False, depth, with_pred, False)
                res[f"v{k}"], res[f"d{k}"] = U(v), UD(dist)
                if with_pred:
                    res[f"p{k}"] = U(pred)
            stats = h.last_traversal_stats()
            out["levels"], out["bottom_up_levels"] = int(stats["steps"]), int(stats["edges_inspected"])
            # multi-source: all roots at once, every rank passing the whole list -- through a SECOND resource handle of the same communicator
            # (the reference accepts any handle: callers that create one per call; the graph's cached traversal plan moves to it)
            h2 = cg.ResourceHandle(comm)
            dist, pred, v = cg.bfs(h2, g, X(roots), False, depth, False, False)
            res["vm"], res["dm"] = U(v), UD(dist)
            del h2
            mine = np.array([roots[0]], np.int32) if rank == 0 else np.zeros(0, np.int32)  # ... and back on the first handle, after the second is gone
            dist, _, v = cg.bfs(h, g, X(mine), False, depth, False, False)
            out["first_handle_again"] = bool(np.array_equal(U(v), res["v0"]) and np.array_equal(UD(dist), res["d0"]))
        else:
            cutoff = float(args[4]) if len(args) > 4 else 3.0e38
            for k, root in enumerate(roots):
                hk = cg.ResourceHandle(comm) if k % 2 == 1 else h  # every other call through a fresh handle of the communicator
                v, dist, pred = cg.sssp(hk, g, xid(root), cutoff, with_pred, False)
                res[f"v{k}"], res[f"d{k}"] = U(v), dist.cpu().numpy()
                if with_pred:
                    res[f"p{k}"] = U(pred)
                del hk
        np.savez(outdir / f"rank{rank}.npz", roots=roots, **res)
        del g
    elif what == "louvain":
        scale = int(args[0])
        from oracle import oracle as orc
        from test_gpu_parity import louvain_rmat_input

        src, dst, w = louvain_rmat_input(orc, scale)  # undirected simple RMAT, integer weights 1..8, both directions, sorted by (src, dst)
        nv = 1 << scale
        mine = np.arange(src.size) % size == rank      # an arbitrary slice: the library routes the edges to their owners
        verts = np.arange(rank, nv, size, dtype=np.int32)
        g = cg.MGGraph(h, cg.GraphProperties(is_symmetric=True), [X(src[mine])], [X(dst[mine])], [T(w[mine])], store_transposed=False, vertices_array=[X(verts)])
        v, c, q = cg.louvain(h, g, 100, 1e-7, 1.0, False)
        np.savez(outdir / f"rank{rank}.npz", v=U(v), c=c.cpu().numpy())
        out["modularity_hex"] = float(q).hex()
        out["rows"] = int(v.numel())
        st = h.last_traversal_stats()
        out["sweeps"] = int(st["steps"])
        del g
    elif what == "props":
        # edge ids / edge type ids on a multi-GPU graph: kept with the rank's slice, back from cugraph_decompress_to_edgelist (graph_mg.cpp:127-151);
        # degrees / has_vertex on the same graph (the wide case sends their id columns through the INT64 translation too)
        scale, id_kind = int(args[0]), args[1]
        (s, d), first = rmat_slice(scale, rank, size)
        nv, ne = 1 << scale, 16 << scale
        w = np.random.default_rng(1).integers(1, 9, size=ne).astype(np.float32)[first: first + s.size].copy()
        ids = ((np.random.default_rng(4).permutation(ne) + 1000).astype(np.int64 if id_kind == "i64" else np.int32))[first: first + s.size].copy()
        types = (np.arange(first, first + s.size) % 5).astype(np.int32)
        verts = np.arange(rank, nv, size, dtype=np.int32)
        g = cg.MGGraph(h, cg.GraphProperties(is_multigraph=True), [X(s)], [X(d)], [T(w)], store_transposed=False, vertices_array=[X(verts)],
                       edge_id_array=[T(ids)], edge_type_array=[T(types)])
        es, ed, ew, ei, et = cg.decompress_to_edgelist(h, g)
        v, din, dout = cg.degrees(h, g, None)
        some = np.arange(rank, nv, 7 * size + 1, dtype=np.int32)  # every rank lists its own few; the owners answer
        v2, din2, dout2 = cg.degrees(h, g, X(some))
        hv = cg.has_vertex(h, g, X(np.array([0, nv - 1, nv + 3], np.int32)))
        np.savez(outdir / f"rank{rank}.npz", s=U(es), d=U(ed), w=ew.cpu().numpy(), ids=ei.cpu().numpy(), types=et.cpu().numpy(), v=U(v), din=din.cpu().numpy(),
                 dout=dout.cpu().numpy(), v2=U(v2), din2=din2.cpu().numpy(), dout2=dout2.cpu().numpy(), listed=some)
        out["has_vertex"] = [bool(x) for x in hv.cpu().numpy()]
        try:  # together with a flag that rewrites the edge list the properties are refused on every rank, as on one GPU
            cg.MGGraph(h, cg.GraphProperties(is_multigraph=True), [X(s)], [X(d)], [T(w)], edge_id_array=[T(ids)], drop_self_loops=True)
            out["refused"] = "accepted"
        except Exception as e:  # noqa: BLE001
            out["refused"] = str(e)
        del g
    elif what == "paths":
        # cugraph_bfs with predecessors + cugraph_extract_paths on a multi-GPU graph: every rank asks for its own destinations, most of them
        # vertices other ranks own
        scale = int(args[0])
        (s, d), first = rmat_slice(scale, rank, size)
        nv = 1 << scale
        verts = np.arange(rank, nv, size, dtype=np.int32)
        g = cg.MGGraph(h, cg.GraphProperties(is_multigraph=True), [X(s)], [X(d)], None, store_transposed=False, vertices_array=[X(verts)])
        all_s, _ = rmat_slice(scale, 0, 1)
        root = int(np.random.default_rng(7).choice(np.unique(all_s[0]), size=1)[0])
        dests = np.random.default_rng(100 + rank).choice(nv, size=40 + 5 * rank, replace=False).astype(np.int32)
        mine = np.array([root], np.int32) if rank == 0 else np.zeros(0, np.int32)
        dist, pred, v, paths = cg.bfs_extract_paths(h, g, X(mine), X(dests))
        np.savez(outdir / f"rank{rank}.npz", root=root, dests=dests, v=U(v), dist=UD(dist), pred=U(pred), paths=U(paths.reshape(-1)).reshape(paths.shape))
        out["width"] = int(paths.shape[1])
        try:  # a destination that is no vertex: INVALID_INPUT on every rank
            cg.bfs_extract_paths(h, g, X(mine), X(np.array([nv + 9], np.int32) if rank == size - 1 else dests[:2]))
            out["bad_destination"] = "accepted"
        except Exception as e:  # noqa: BLE001
            out["bad_destination"] = str(e)
        del g
    else:
        raise ValueError(what)
    return out
