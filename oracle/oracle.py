"""ctypes front-end of the CPU oracle (oracle/oracle.c) and, when built, of the reference's own CPU
reference functions compiled in place (oracle/_ref/libref.so, recipe oracle/build_ref.sh).

TEST INFRASTRUCTURE ONLY: the product path (cugraph_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = _HERE / "_build" / "liboracle.so"
_REF = _HERE / "_ref" / "libref.so"

INT32_MAX = np.iinfo(np.int32).max
FLT_MAX = np.finfo(np.float32).max
DBL_MAX = np.finfo(np.float64).max


def build(force: bool = False) -> None:
    """Compile liboracle.so (gcc) and, if /root/reference is present, libref.so."""
    if force or not _LIB.exists() or _LIB.stat().st_mtime < (_HERE / "oracle.c").stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "-s"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if os.path.isdir(os.environ.get("CUGRAPH_REFERENCE_DIR", "/root/reference")):
        if force or not _REF.exists():
            subprocess.check_call(["bash", str(_HERE / "build_ref.sh")], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_LIB))
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def ref():
    """The reference's own compiled CPU functions, or None when libref.so was never built."""
    global _ref
    if _ref is None and _REF.exists():
        _ref = C.CDLL(str(_REF))
    return _ref


def num_threads() -> int:
    return int(lib().orc_num_threads())


def _p(a, t=None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


# ------------------------------------------------------------------------------------------ RMAT
def rmat(scale: int, num_edges: int, a=0.57, b=0.19, c=0.19, seed=0, first_edge=0):
    src = np.empty(num_edges, np.int32)
    dst = np.empty(num_edges, np.int32)
    lib().orc_rmat(C.c_int(scale), C.c_uint64(first_edge), C.c_uint64(num_edges), C.c_double(a),
                   C.c_double(b), C.c_double(c), C.c_uint64(seed), _p(src), _p(dst))
    return src, dst


# ------------------------------------------------------------------------------- COO -> CSR/CSC
def coo_to_cs(nv: int, major, minor, weights=None):
    """Compressed sparse by `major`, neighbours ascending; returns (offsets int64, indices int32, w)."""
    major, minor = _i32(major), _i32(minor)
    ne = major.size
    offsets = np.empty(nv + 1, np.int64)
    indices = np.empty(ne, np.int32)
    out_w = None
    wsize = 0
    if weights is not None:
        weights = np.ascontiguousarray(weights)
        assert weights.dtype in (np.float32, np.float64)
        out_w = np.empty(ne, weights.dtype)
        wsize = weights.dtype.itemsize
    lib().orc_coo_to_cs(C.c_int64(nv), C.c_int64(ne), _p(major), _p(minor), _p(weights), C.c_int(wsize),
                        _p(offsets), _p(indices), _p(out_w))
    return offsets, indices, out_w


# ----------------------------------------------------------------------------- hypersparse rows
# Restatement of compress_hypersparse_offsets (cpp/src/structure/detail/structure_utils.cuh:139-195): the CSR + DCSR hybrid.  Rows below
# `first` keep one offset each; of the rows >= first only those with an edge survive: `nzd` lists them in ascending order (the reference's
# dcs_nzd_vertices) and the returned offsets have first + len(nzd) + 1 entries, the last one being the edge count.  Parity status: pinned by the
# definition only (the reference's tests hold no vector for it) plus the known-answer case in tests/test_oracle.py; the lookup the device view
# performs (edge_partition_device_view.cuh:43-58: lower_bound over dcs_nzd_vertices) is hypersparse_find below.
def compress_hypersparse_offsets(offsets, first):
    offsets = np.asarray(offsets, dtype=np.int64)
    nv = len(offsets) - 1
    first = max(0, min(int(first), nv))
    deg = np.diff(offsets[first:])
    nzd = (first + np.nonzero(deg > 0)[0]).astype(np.int32)
    out = np.concatenate([offsets[:first], offsets[nzd], offsets[-1:]])
    return out.astype(np.int64), nzd


def hypersparse_find(nzd, first, n_stored, row):
    """stored index of `row` in the hybrid form, or -1 when the row is not stored (major_hypersparse_idx_from_major_nocheck)"""
    if row < first:
        return row if row < n_stored else -1
    k = int(np.searchsorted(nzd, row, side="left"))
    return first + k if k < len(nzd) and nzd[k] == row else -1


def inflate_hypersparse_offsets(offsets, nzd, first, nv):
    """the plain offsets [nv + 1] back from the hybrid form"""
    offsets = np.asarray(offsets, dtype=np.int64)
    deg = np.zeros(nv, np.int64)
    stored_deg = np.diff(offsets)
    deg[:first] = stored_deg[:first]
    deg[np.asarray(nzd, dtype=np.int64)] = stored_deg[first:]
    return np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)


# -------------------------------------------------------------------------------------- PageRank
def pagerank(nv, offsets, indices, weights=None, alpha=0.85, epsilon=1e-6, max_iter=100,
             personalization=None, initial_guess=None, precomputed_outw=None, acc64=True,
             dtype=np.float32):
    """CSC input (row = destination).  Returns (pr, iterations, converged)."""
    dtype = np.dtype(dtype)
    fn = lib().orc_pagerank_f32 if dtype == np.float32 else lib().orc_pagerank_f64
    fn.restype = C.c_int
    offsets, indices = _i64(offsets), _i32(indices)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=dtype)
    ow = None if precomputed_outw is None else np.ascontiguousarray(precomputed_outw, dtype=dtype)
    if personalization is not None:
        pv, pval = _i32(personalization[0]), np.ascontiguousarray(personalization[1], dtype=dtype)
        n_pers = pv.size
    else:
        pv = pval = None
        n_pers = 0
    if initial_guess is not None:
        pr = np.array(initial_guess, dtype=dtype, copy=True)
    else:
        pr = np.empty(nv, dtype)
    iters = C.c_int64(0)
    conv = fn(C.c_int64(nv), _p(offsets), _p(indices), _p(w), _p(ow), C.c_int64(n_pers), _p(pv), _p(pval),
              C.c_int(initial_guess is not None), C.c_double(alpha), C.c_double(epsilon),
              C.c_int64(max_iter), C.c_int(1 if acc64 else 0), _p(pr), C.byref(iters))
    return pr, int(iters.value), bool(conv)


# ------------------------------------------------------------------------------------------- BFS
def bfs(nv, offsets, indices, sources, depth_limit=INT32_MAX):
    """CSR input (row = source).  Returns (dist int32, pred int32) with first-discoverer parents."""
    offsets, indices, sources = _i64(offsets), _i32(indices), _i32(np.atleast_1d(sources))
    dist = np.empty(nv, np.int32)
    pred = np.empty(nv, np.int32)
    lib().orc_bfs(C.c_int64(nv), _p(offsets), _p(indices), _p(sources), C.c_int64(sources.size),
                  C.c_int64(depth_limit), _p(dist), _p(pred))
    return dist, pred


def bfs_min_pred(nv, offsets, indices, dist):
    offsets, indices, dist = _i64(offsets), _i32(indices), _i32(dist)
    pred = np.empty(nv, np.int32)
    lib().orc_bfs_min_pred(C.c_int64(nv), _p(offsets), _p(indices), _p(dist), _p(pred))
    return pred


# ------------------------------------------------------------------------------------------ SSSP
def sssp(nv, offsets, indices, weights, source, cutoff=np.inf):
    weights = np.ascontiguousarray(weights)
    dt = weights.dtype
    fn = lib().orc_sssp_f32 if dt == np.float32 else lib().orc_sssp_f64
    offsets, indices = _i64(offsets), _i32(indices)
    dist = np.empty(nv, dt)
    pred = np.empty(nv, np.int32)
    cut = float(cutoff) if np.isfinite(cutoff) else float(DBL_MAX)
    fn(C.c_int64(nv), _p(offsets), _p(indices), _p(weights), C.c_int32(int(source)), C.c_double(cut),
       _p(dist), _p(pred))
    return dist, pred


def sssp_min_pred(nv, offsets, indices, weights, source, dist):
    weights = np.ascontiguousarray(weights)
    dt = weights.dtype
    fn = lib().orc_sssp_min_pred_f32 if dt == np.float32 else lib().orc_sssp_min_pred_f64
    offsets, indices = _i64(offsets), _i32(indices)
    dist = np.ascontiguousarray(dist, dtype=dt)
    pred = np.empty(nv, np.int32)
    fn(C.c_int64(nv), _p(offsets), _p(indices), _p(weights), C.c_int32(int(source)), _p(dist), _p(pred))
    return pred


# --------------------------------------------------- the reference's own functions (oracle/_ref)
def ref_pagerank(nv, offsets, indices, weights=None, alpha=0.85, epsilon=1e-6, max_iter=100,
                 personalization=None, initial_guess=None, dtype=np.float32):
    """pagerank_reference (cpp/tests/link_analysis/pagerank_test.cpp:33-121) itself."""
    r = ref()
    assert r is not None, "oracle/_ref/libref.so not built"
    dtype = np.dtype(dtype)
    f32 = dtype == np.float32
    fn = r.ref_pagerank_f32 if f32 else r.ref_pagerank_f64
    fl = C.c_float if f32 else C.c_double
    offsets, indices = _i64(offsets), _i32(indices)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=dtype)
    if personalization is not None:
        pv, pval = _i32(personalization[0]), np.ascontiguousarray(personalization[1], dtype=dtype)
        n_pers = pv.size
    else:
        pv = pval = None
        n_pers = 0
    pr = np.array(initial_guess, dtype=dtype, copy=True) if initial_guess is not None else np.zeros(nv, dtype)
    failed = fn(C.c_int64(nv), _p(offsets), _p(indices), _p(w), C.c_int64(n_pers), _p(pv), _p(pval), _p(pr),
                fl(alpha), fl(epsilon), C.c_int64(max_iter), C.c_int(initial_guess is not None))
    return pr, bool(failed)


def ref_bfs(nv, offsets, indices, source, depth_limit=INT32_MAX):
    r = ref()
    assert r is not None
    offsets, indices = _i64(offsets), _i32(indices)
    dist = np.empty(nv, np.int32)
    pred = np.empty(nv, np.int32)
    r.ref_bfs(C.c_int64(nv), _p(offsets), _p(indices), _p(dist), _p(pred), C.c_int32(int(source)),
              C.c_int32(int(depth_limit)))
    return dist, pred


def ref_sssp(nv, offsets, indices, weights, source, cutoff=None):
    r = ref()
    assert r is not None
    weights = np.ascontiguousarray(weights)
    dt = weights.dtype
    f32 = dt == np.float32
    fn = r.ref_sssp_f32 if f32 else r.ref_sssp_f64
    fl = C.c_float if f32 else C.c_double
    offsets, indices = _i64(offsets), _i32(indices)
    dist = np.empty(nv, dt)
    pred = np.empty(nv, np.int32)
    cut = (FLT_MAX if f32 else DBL_MAX) if cutoff is None else cutoff
    fn(C.c_int64(nv), _p(offsets), _p(indices), _p(weights), _p(dist), _p(pred), C.c_int32(int(source)), fl(cut))
    return dist, pred


# ------------------------------------------------------------------------- edge-list preprocessing
# CPU restatement of the graph-creation flags (cpp/src/c_api/graph_sg.cpp:185-248).  Parity status: UNPINNED -- the
# reference's C-API tests hold no golden vectors for these flags; the restatement follows the documented semantics
# (graph_functions.hpp:420-466, 1073-1140) and the operator of symmetrize_edgelist_impl.cuh:60-135.
def remove_self_loops(src, dst, w=None):
    keep = src != dst
    return src[keep], dst[keep], (None if w is None else w[keep])


def remove_multi_edges(src, dst, w=None):
    """One edge per (src, dst); with weights the minimum-weight one (keep_min_value_edge; a valid 'arbitrary' pick too)."""
    if w is None:
        order = np.lexsort((dst, src))
    else:
        order = np.lexsort((w, dst, src))
    s, d = src[order], dst[order]
    first = np.ones(s.size, bool)
    first[1:] = (s[1:] != s[:-1]) | (d[1:] != d[:-1])
    return s[first], d[first], (None if w is None else w[order][first])


def symmetrize_edgelist(src, dst, w=None):
    """reciprocal = False: self-loops kept; per unordered pair the lower (src > dst) and upper (src < dst) edges are sorted
    by weight and paired rank by rank -- matched pairs become one edge with the average weight, unmatched edges keep
    theirs -- and every resulting edge appears in both directions (symmetrize_edgelist_impl.cuh:78-110, 949-960)."""
    diag = src == dst
    ds, dd = src[diag], dst[diag]
    dw = None if w is None else w[diag]
    s, d = src[~diag], dst[~diag]
    ww = None if w is None else w[~diag]
    hi, lo = np.maximum(s, d), np.minimum(s, d)
    upper = (s < d).astype(np.int8)
    order = np.lexsort((upper, lo, hi)) if ww is None else np.lexsort((ww, upper, lo, hi))
    hi, lo, upper = hi[order], lo[order], upper[order]
    ww = None if ww is None else ww[order]
    out_hi, out_lo, out_w = [], [], []
    n, i = hi.size, 0
    while i < n:
        j = i
        while j < n and hi[j] == hi[i] and lo[j] == lo[i]:
            j += 1
        nlo = int((upper[i:j] == 0).sum())
        nup = (j - i) - nlo
        for k in range(max(nlo, nup)):
            out_hi.append(hi[i]); out_lo.append(lo[i])
            if ww is not None:
                if k < nlo and k < nup:
                    out_w.append((ww[i + k] + ww[i + nlo + k]) / ww.dtype.type(2))
                elif k < nlo:
                    out_w.append(ww[i + k])
                else:
                    out_w.append(ww[i + nlo + k])
        i = j
    oh, ol = np.array(out_hi, src.dtype), np.array(out_lo, src.dtype)
    rs = np.concatenate([oh, ol, ds]); rd = np.concatenate([ol, oh, dd])
    rw = None if w is None else np.concatenate([np.array(out_w, w.dtype), np.array(out_w, w.dtype), dw])
    return rs, rd, rw


# ----------------------------------------------------------------------------------------- Louvain
# Restatement of cugraph::louvain with rng_state = nullopt (cpp/src/community/louvain_impl.cuh:40-287) and of its
# helpers (cpp/src/community/detail/common_methods.cuh:52-479): synchronous local moving with the up/down rule, modularity
# test per sweep, graph contraction per level, dendrogram flattening.  Arithmetic in fp64, every formula written in the
# operation order of the reference so that the HIP path (which evaluates the same expressions without contraction) takes
# the same decisions.  Cluster LABELS of a contracted level are the rank of the old label among the labels in use
# (ascending); the reference takes the vertex ids its coarsen_graph renumbering happens to assign (degree order, unstable
# sort on ties), so labels are comparable only up to a renaming -- the tests compare partitions and the modularity.
# Pinned by cpp/tests/c_api/louvain_test.c: test_louvain (Q = 0.215969, {0,1,2},{3,4,5}) and test_louvain_no_weight
# (Q = 0.125, {0,1,2,3},{4,5}).
def louvain_modularity(src, dst, w, clusters, resolution=1.0):
    """Q of a clustering, as detail::compute_modularity (common_methods.cuh:176-228)."""
    src, dst = np.asarray(src, np.int64), np.asarray(dst, np.int64)
    w = np.asarray(w, np.float64)
    c = np.asarray(clusters, np.int64)
    m = w.sum()
    k = np.bincount(src, weights=w, minlength=c.size)
    a = np.bincount(c, weights=k, minlength=int(c.max()) + 1 if c.size else 1)
    internal = w[c[src] == c[dst]].sum()
    return internal / m - (resolution * (a * a).sum()) / (m * m)


def _louvain_level(nv, src, dst, w, threshold, resolution, m, noise_floor=1e-15):
    """One level: returns (clusters of the level's vertices, Q reached)."""
    k = np.zeros(nv)
    np.add.at(k, src, w)                                     # vertex weights (out-weight sums)
    c = np.arange(nv, dtype=np.int64)                        # every vertex its own cluster
    accepted = c.copy()

    def q_of(cl, a):
        internal = w[cl[src] == cl[dst]].sum()
        return internal / m - (resolution * (a * a).sum()) / (m * m)

    a = k.copy()                                             # cluster weights, indexed by cluster label
    new_q = q_of(c, a)
    cur_q = new_q - 1.0
    up_down = True
    min_gain = max(threshold / max(nv, 1), noise_floor)      # compute_louvain_min_vertex_move_gain (common_methods.cuh:60-66)
    order = np.arange(src.size)
    while new_q > cur_q + threshold:
        cur_q = new_q
        # aggregated weight from every vertex to every neighbouring cluster (edges in input order inside a pair)
        cd = c[dst]
        o = np.lexsort((order, cd, src))
        s_s, s_c, s_w, s_d = src[o], cd[o], w[o], dst[o]
        head = np.ones(s_s.size, bool)
        head[1:] = (s_s[1:] != s_s[:-1]) | (s_c[1:] != s_c[:-1])
        starts = np.flatnonzero(head)
        seg_v, seg_c = s_s[starts], s_c[starts]
        seg_id = np.cumsum(head) - 1
        seg_sum = np.zeros(starts.size)
        np.add.at(seg_sum, seg_id, s_w)                      # sequential, in sorted order (reduceat sums pairwise)
        # old_cluster_sum (same cluster, not a self-loop) and cluster_subtract (self-loops) per vertex
        old_sum = np.zeros(nv)
        sub = np.zeros(nv)
        same = (s_c == c[s_s])
        loop = (s_d == s_s)
        np.add.at(old_sum, s_s[same & ~loop], s_w[same & ~loop])
        np.add.at(sub, s_s[loop], s_w[loop])
        kk = k[seg_v]
        new_sum = np.where(seg_c == c[seg_v], seg_sum - sub[seg_v], seg_sum)
        a_new, a_old = a[seg_c], a[c[seg_v]]
        delta = 2.0 * (((new_sum - old_sum[seg_v]) / m) - resolution * (a_new * kk - a_old * kk + kk * kk) / (m * m))
        best_c = np.full(nv, -1, np.int64)
        best_d = np.zeros(nv)
        for v, cc, dd in zip(seg_v.tolist(), seg_c.tolist(), delta.tolist()):  # segments ascend in (vertex, cluster)
            if dd > best_d[v]:
                best_d[v], best_c[v] = dd, cc
        want = best_d > min_gain
        moves = want & ((best_c > c) == up_down)
        if not moves.any():  # update_clustering_by_delta_modularity takes up_down BY VALUE (common_methods.cuh:277): its flip lasts for this sweep only
            moves = want & ((best_c > c) == (not up_down))
        c = np.where(moves, best_c, c)
        a = np.zeros(nv)
        np.add.at(a, c, k)
        up_down = not up_down
        new_q = q_of(c, a)
        if new_q > cur_q + threshold:
            accepted = c.copy()
    return accepted, cur_q


def louvain_noise_floor(w):
    """louvain_delta_modularity_noise_floor (common_methods.cuh:52-58): 1e-12 for a float graph, 1e-15 for a double one.
    A graph without weights, or with float32 weights, is a float graph at the C API (graph_sg.cpp:775-779)."""
    return 1e-15 if (w is not None and np.asarray(w).dtype == np.float64) else 1e-12


def louvain(nv, src, dst, w=None, max_level=100, threshold=1e-7, resolution=1.0, noise_floor=None):
    """Returns (clusters per vertex, modularity, levels).  Graph = directed edge list (an undirected graph lists both
    directions, as the C API's symmetric graphs do); w = None means weight 1."""
    if noise_floor is None:
        noise_floor = louvain_noise_floor(w)
    src, dst = np.asarray(src, np.int64), np.asarray(dst, np.int64)
    w = np.ones(src.size) if w is None else np.asarray(w, np.float64)
    m = w.sum()
    part = np.arange(nv, dtype=np.int64)
    best = -1.0
    levels = 0
    cur_nv = nv
    while levels < max_level:
        levels += 1
        c, q = _louvain_level(cur_nv, src, dst, w, threshold, resolution, m, noise_floor)
        if q <= best:
            break
        best = q
        # graph_contraction (common_methods.cuh:231-263): coarsen_graph(..., renumber = true) numbers the coarse vertices as every graph
        # creation does -- by degree, descending (renumber_edgelist_impl.cuh: "4. sort local vertices by degree (descending)", a stable key
        # sort over the id-sorted vertex list, so equal degrees keep ascending label order); degree = number of coarse edges leaving the
        # vertex, i.e. of DISTINCT neighbour clusters (its own included when it has internal weight).  The ids decide every tie of the next
        # level (smaller cluster id wins, reduce_op_t) and its up / down rule, so they are part of the algorithm: with label-order ids the
        # restatement missed the reference's karate goldens (cpp/tests/community/louvain_test.cpp:228-237) at resolution 1.
        used = np.zeros(cur_nv, bool)
        used[c] = True
        labels = np.flatnonzero(used)
        pair_keys = np.unique(c[src].astype(np.int64) * cur_nv + c[dst])
        deg = np.bincount(pair_keys // cur_nv, minlength=cur_nv)[labels]
        by_degree = np.lexsort((labels, -deg))
        rank = np.full(cur_nv, -1, np.int64)
        rank[labels[by_degree]] = np.arange(labels.size)
        c = rank[c]
        part = c[part]
        cur_nv = int(used.sum())
        cs, cd = c[src], c[dst]
        o = np.lexsort((np.arange(src.size), cd, cs))
        cs, cd, ww = cs[o], cd[o], w[o]
        head = np.ones(cs.size, bool)
        head[1:] = (cs[1:] != cs[:-1]) | (cd[1:] != cd[:-1])
        starts = np.flatnonzero(head)
        src, dst = cs[starts], cd[starts]
        w = np.zeros(starts.size)
        np.add.at(w, np.cumsum(head) - 1, ww)
    return part.astype(np.int32), float(best), levels


def louvain_c(nv, src, dst, w=None, max_level=100, threshold=1e-7, resolution=1.0, noise_floor=None):
    """oracle.c: orc_louvain -- the same algorithm as louvain() above, in C, for graphs beyond the numpy version's reach
    (tests/test_oracle.py checks the two against each other).  Returns (clusters, modularity, levels, sweeps)."""
    src, dst = _i32(src), _i32(dst)
    if src.size and np.any(src[1:] < src[:-1]):  # edges must be grouped by source; a stable sort keeps the order inside a source
        o = np.argsort(src, kind="stable")
        src, dst = np.ascontiguousarray(src[o]), np.ascontiguousarray(dst[o])
        w = None if w is None else np.asarray(w)[o]
    wd = None if w is None else np.ascontiguousarray(w, np.float64)
    clusters = np.empty(nv, np.int32)
    q = C.c_double(0.0)
    sweeps = C.c_int(0)
    lib().orc_louvain_set_noise_floor(C.c_double(louvain_noise_floor(w) if noise_floor is None else noise_floor))
    fn = lib().orc_louvain
    fn.restype = C.c_int
    levels = fn(C.c_int64(nv), C.c_int64(src.size), _p(src), _p(dst), None if wd is None else _p(wd), C.c_int64(max_level), C.c_double(threshold),
                C.c_double(resolution), _p(clusters), C.byref(q), C.byref(sweeps))
    return clusters, float(q.value), int(levels), int(sweeps.value)

