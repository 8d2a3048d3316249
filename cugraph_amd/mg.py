"""Multi-GPU PageRank: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).

Partitioning (re-designed for a full-mesh xGMI node instead of translated):
  * vertices are ordered by descending GLOBAL in-degree (ties: ascending id) and dealt round-robin to the P ranks:
    position p -> owner p % P, local row p // P.  Every rank gets the same mix of hub and tail rows (balanced edge
    counts) and its local rows are already degree-sorted;
  * 1-D by destination: the owner of a destination holds ALL its in-edges, so the pull-SpMV needs no partial-sum
    reduction over ranks (the reference's 2-D scheme needs a row broadcast AND a column reduce per iteration,
    prims/update_edge_src_dst_property.cuh:550-579 + prims/detail/per_v_transform_reduce_e.cuh:3390-3406);
  * per iteration ONE collective: a SPARSE all-to-all of x = pr / out_w.  A rank receives exactly the source values
    its edges reference -- vertices without out-edges (most of an RMAT graph) are never sent, a low-degree source only
    goes where one of its out-edges lives -- as one point-to-point message per peer, so all xGMI links of a GPU work at
    once; a ring all-gather of the whole vector would be bound by one link and move (P-1)/P * 4V bytes into every GPU.
    The iteration's scalars (partial L1 change, partial dangling mass, max |x|) ride in a 32-byte tail of every message
    -- no scalar all-reduce, every rank adds the P partials in rank order (deterministic).
Local column ids are compact (the distinct sources a rank references, hottest first), so the column-tiled SpMV kernels
run unchanged on the unpacked receive buffer.

The local compute sits behind `LocalEngine`; the product engine is `HipLocalEngine` (C ABI, HIP).  Tests plug a
CPU engine built on the oracle into the same orchestration to exercise partitioning + collectives under gloo.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import time

import torch
import torch.distributed as dist

TAIL_BYTES = 32  # (L1 change, dangling mass, max |x|) as 3 doubles + padding, behind every message


# ------------------------------------------------------------------------------- device primitives
# Plan construction on the GPU uses the LIBRARY's stable radix sort and exclusive scan through the C ABI
# (cugraph_amd_sort_pairs_u64_u32 / cugraph_amd_exclusive_scan_u32), not torch.sort / argsort / unique / cumsum (rocPRIM under
# the hood): the product's multi-GPU path runs no Thrust / CUB / rocPRIM kernel.  Host tensors (the CPU engines of the gloo
# tests) take the torch functions.
_PRIM = {}


def _prim_handle():
    h = _PRIM.get("handle")
    if h is None:
        from .pylib import ResourceHandle

        h = _PRIM["handle"] = ResourceHandle()
    return h


def release_build_temporaries():
    """After a partition / plan has been built: hand the library's large cached blocks (sort buffers of the construction) back to the
    driver.  torch and RCCL allocate from the same device (exchange buffers, a second edge-list copy for the bottom-up BFS) and could hit
    out-of-memory while the library's pool sits on tens of GB of idle cache (round-3 review)."""
    from . import _capi as capi

    return int(capi.lib().cugraph_amd_memory_pool_trim_large(256 << 20))


def _bits_for(max_value: int) -> int:
    return max(1, int(max_value).bit_length())


def _lib_sort_pairs(keys: torch.Tensor, vals: torch.Tensor, bits: int):
    """in place: keys (int64, non-negative) ascending on their low `bits` bits, stable; vals (int32) follow"""
    from . import _capi as capi
    from .pylib import assert_success

    assert keys.is_cuda and keys.dtype == torch.int64 and keys.is_contiguous() and vals.dtype == torch.int32 and vals.is_contiguous()
    torch.cuda.current_stream().synchronize()  # the library sorts on its own stream
    err = C.c_void_p()
    code = capi.lib().cugraph_amd_sort_pairs_u64_u32(_prim_handle().c_resource_handle_ptr, C.c_void_p(keys.data_ptr()), C.c_void_p(vals.data_ptr()),
                                                     keys.numel(), 0, int(bits), C.byref(err))
    assert_success(code, err, "cugraph_amd_sort_pairs_u64_u32")


def _lib_exclusive_scan(flags: torch.Tensor) -> torch.Tensor:
    from . import _capi as capi
    from .pylib import assert_success

    assert flags.is_cuda and flags.dtype == torch.int32 and flags.is_contiguous()
    out = torch.empty_like(flags)
    torch.cuda.current_stream().synchronize()
    err = C.c_void_p()
    code = capi.lib().cugraph_amd_exclusive_scan_u32(_prim_handle().c_resource_handle_ptr, C.c_void_p(flags.data_ptr()), C.c_void_p(out.data_ptr()), flags.numel(),
                                                     C.byref(err))
    assert_success(code, err, "cugraph_amd_exclusive_scan_u32")
    return out


def stable_argsort(x: torch.Tensor, max_value=None) -> torch.Tensor:
    """positions that sort the non-negative integers x ascending, ties in input order (int64 positions)"""
    if not x.is_cuda:
        return torch.argsort(x, stable=True)
    n = x.numel()
    if n == 0:
        return torch.empty(0, dtype=torch.int64, device=x.device)
    keys = x.to(torch.int64).contiguous().clone()
    vals = torch.arange(n, dtype=torch.int32, device=x.device)
    _lib_sort_pairs(keys, vals, _bits_for(int(x.max()) if max_value is None else max_value))
    return vals.to(torch.int64)


def inclusive_counts(rows: torch.Tensor, n_rows: int) -> torch.Tensor:
    """out[r] = number of entries of `rows` that are <= r (the CSR offsets of a row-sorted list, without their leading 0)"""
    cnt = torch.bincount(rows, minlength=n_rows)
    if not rows.is_cuda:
        return torch.cumsum(cnt, 0)
    c32 = torch.cat([cnt.to(torch.int32), torch.zeros(1, dtype=torch.int32, device=rows.device)]).contiguous()
    return _lib_exclusive_scan(c32)[1:].to(torch.int64)


def degree_order(deg: torch.Tensor) -> torch.Tensor:
    """vertices by descending degree, ties by ascending id (= torch.sort(deg, descending=True, stable=True).indices)"""
    if not deg.is_cuda:
        return torch.sort(deg, descending=True, stable=True)[1]
    mx = int(deg.max()) if deg.numel() else 0
    return stable_argsort(mx - deg.to(torch.int64), mx)


def unique_inverse(x: torch.Tensor):
    """(sorted distinct values of the non-negative integers x, index of every element's value in that list)
    = torch.unique(x, sorted=True, return_inverse=True)"""
    if not x.is_cuda:
        return torch.unique(x, sorted=True, return_inverse=True)
    n = x.numel()
    if n == 0:
        return x.clone(), torch.empty(0, dtype=torch.int64, device=x.device)
    keys = x.to(torch.int64).contiguous().clone()
    vals = torch.arange(n, dtype=torch.int32, device=x.device)
    _lib_sort_pairs(keys, vals, _bits_for(int(x.max())))
    head = torch.ones(n, dtype=torch.int32, device=x.device)
    head[1:] = (keys[1:] != keys[:-1]).to(torch.int32)
    rank = (_lib_exclusive_scan(head) + head - 1).to(torch.int64)  # index of the element's value among the distinct values
    n_unique = int(rank[-1]) + 1
    uniq = torch.empty(n_unique, dtype=x.dtype, device=x.device)
    uniq[rank] = keys.to(x.dtype)            # every member of a group stores the group's value
    inverse = torch.empty(n, dtype=torch.int64, device=x.device)
    inverse[vals.to(torch.int64)] = rank
    return uniq, inverse


def _share_stream(handle, dev):
    """A stream of torch's that the library borrows (cugraph_amd_handle_set_stream): the collectives of an iteration are ordered on
    torch's CURRENT stream, so stepping runs with this stream current (on_engine_stream) and the kernels that consume / produce the
    exchange buffers need no host synchronisation.  Never torch's default stream: its handle is the null stream, which
    cugraph_amd_handle_set_stream reads as "back to the handle's own stream" (the library would then race the collectives)."""
    s = torch.cuda.Stream(device=dev)
    assert s.cuda_stream != 0
    s.wait_stream(torch.cuda.current_stream())
    handle.set_stream(s.cuda_stream)
    return s


@contextlib.contextmanager
def on_engine_stream(engine):
    """Makes the engine's borrowed stream torch's current one (ordered behind / ahead of the caller's stream at entry / exit)."""
    s = getattr(engine, "stream", None)
    if s is None:
        yield
        return
    outer = torch.cuda.current_stream()
    s.wait_stream(outer)
    with torch.cuda.stream(s):
        yield
    outer.wait_stream(s)


def _order_engine_stream_behind_caller(engine):
    s = getattr(engine, "stream", None)
    if s is not None:
        s.wait_stream(torch.cuda.current_stream())


# ------------------------------------------------------------------------------------------ partition
class Partition:
    """Global degree-order numbering dealt round-robin over the ranks."""

    def __init__(self, in_degree: torch.Tensor, world: int, rank: int):
        self.nv = int(in_degree.numel())
        self.world, self.rank = world, rank
        # stable descending sort: ties keep ascending vertex id
        order = degree_order(in_degree)
        self.order = order                                   # position -> vertex
        self.pos = torch.empty_like(order)
        self.pos[order] = torch.arange(self.nv, dtype=order.dtype, device=order.device)  # vertex -> position
        self.local_vertices = order[rank::world]             # external ids of the rows this rank owns, local order
        self.n_rows = int(self.local_vertices.numel())


def _count_owners(owner: torch.Tensor, world: int) -> torch.Tensor:
    """Elements per owner rank.  Not torch.bincount: its histogram kernel dies with SIGFPE for ~10^9 elements and a handful
    of bins (seen on ROCm 7 / torch 2.10 at RMAT-26 with one rank); `world` compare-and-sum passes are cheap."""
    return torch.stack([(owner == r).sum() for r in range(world)]).to(torch.int64)


_A2A_MAX_BYTES = 1 << 30  # per message and round


def _a2a(t, send_counts, recv_counts, group):
    """all-to-all-v of a 1-D tensor.  Messages are cut into rounds of at most 1 GiB: a single 8.6 GB message (RMAT-26 edge
    list, one rank) came back truncated from all_to_all_single on ROCm 7 / torch 2.10 -- the one-rank 'exchange' then produced a
    graph with 20 % fewer distinct sources, silently.  One rank needs no collective at all."""
    world = len(send_counts)
    if world == 1:
        return t.clone()
    out = torch.empty(int(sum(recv_counts)), dtype=t.dtype, device=t.device)
    lim = max(1, _A2A_MAX_BYTES // t.element_size())
    biggest = max(max(send_counts), max(recv_counts), 0)
    # every rank must run the same number of rounds: agree on the largest message anywhere
    big_t = torch.tensor([biggest], dtype=torch.int64, device=t.device if dist.get_backend(group) != "gloo" else "cpu")
    dist.all_reduce(big_t, op=dist.ReduceOp.MAX, group=group)
    rounds = max(1, -(-int(big_t.item()) // lim))
    if rounds == 1:
        dist.all_to_all_single(out, t, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts), group=group)
        return out
    s_off = [0] * world
    r_off = [0] * world
    for p in range(1, world):
        s_off[p] = s_off[p - 1] + send_counts[p - 1]
        r_off[p] = r_off[p - 1] + recv_counts[p - 1]
    for k in range(rounds):
        sc = [max(0, min(lim, send_counts[p] - k * lim)) for p in range(world)]
        rc = [max(0, min(lim, recv_counts[p] - k * lim)) for p in range(world)]
        send = torch.cat([t[s_off[p] + k * lim: s_off[p] + k * lim + sc[p]] for p in range(world)]) if sum(sc) else t[:0]
        recv = torch.empty(int(sum(rc)), dtype=t.dtype, device=t.device)
        dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=sc, group=group)
        at = 0
        for p in range(world):
            out[r_off[p] + k * lim: r_off[p] + k * lim + rc[p]] = recv[at: at + rc[p]]
            at += rc[p]
    return out


def _exchange_edges(col_src, local_dst, owner_dst, weights, world, group):
    """Routes every edge to the owner of its destination (all-to-all-v), like shuffle_ext_edges in the reference
    (cpp/src/c_api/graph_mg.cpp:140) but keyed on the degree-order owner instead of a hash."""
    order = stable_argsort(owner_dst, world - 1)
    col_src, local_dst = col_src[order].contiguous(), local_dst[order].contiguous()
    if weights is not None:
        weights = weights[order].contiguous()
    send_counts = _count_owners(owner_dst, world)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    sc, rc = send_counts.tolist(), recv_counts.tolist()
    return (_a2a(col_src, sc, rc, group), _a2a(local_dst, sc, rc, group), (_a2a(weights, sc, rc, group) if weights is not None else None))


class Exchange:
    """Static plan of the per-iteration sparse all-to-all of x (built once, collectively).

    need          sorted global positions of the distinct sources this rank's edges reference (= compact column c -> position)
    recv_counts   values received from each rank;  send_counts: values sent to each rank
    send_index    local row (of this rank) of the k-th value sent, grouped by destination rank
    col_pos       element offset in the receive buffer of compact column c
    """

    def __init__(self, src_pos: torch.Tensor, world: int, rank: int, itemsize: int, group):
        self.world, self.rank = world, rank
        self.tail = TAIL_BYTES // itemsize
        need, self.col_of_edge = unique_inverse(src_pos)
        self.ncols = int(need.numel())
        owner = need % world
        order = stable_argsort(owner, world - 1)             # requests grouped by owner, ascending position inside
        req = (need // world)[order].to(torch.int32)
        counts = _count_owners(owner, world).tolist()
        # fp32 tails hold doubles: keep every message 8-byte aligned by padding odd requests with a repeat of local row 0
        pad_to = 8 // itemsize
        pieces, rc, first = [], [], 0
        for s in range(world):
            piece = req[first:first + counts[s]]
            first += counts[s]
            extra = (-counts[s]) % pad_to
            if extra:
                piece = torch.cat([piece, torch.zeros(extra, dtype=torch.int32, device=piece.device)])
            pieces.append(piece)
            rc.append(int(piece.numel()))
        req_padded = torch.cat(pieces) if pieces else req
        self.recv_counts = rc
        # element offset of every compact column in the receive buffer (messages are followed by their tails)
        seg_start, off = [], 0
        for s in range(world):
            seg_start.append(off)
            off += rc[s] + self.tail
        self.recv_elems = off
        pos_in_group = torch.empty(self.ncols, dtype=torch.int64, device=need.device)
        first = 0
        for s in range(world):
            pos_in_group[first:first + counts[s]] = torch.arange(counts[s], device=need.device) + seg_start[s]
            first += counts[s]
        col_pos = torch.empty(self.ncols, dtype=torch.int64, device=need.device)
        col_pos[order] = pos_in_group
        self.col_pos = col_pos.to(torch.int32)
        # tell every owner what we need from it
        rc_t = torch.tensor(rc, dtype=torch.int64, device=need.device)
        sc_t = torch.empty_like(rc_t)
        dist.all_to_all_single(sc_t, rc_t, group=group)
        self.send_counts = sc_t.tolist()
        self.send_index = _a2a(req_padded, rc, self.send_counts, group)  # what the others asked of us
        self.send_elems = int(sum(self.send_counts)) + world * self.tail
        self.send_splits = [c + self.tail for c in self.send_counts]
        self.recv_splits = [c + self.tail for c in self.recv_counts]


# --------------------------------------------------------------------------------------------- engines
class LocalEngine:
    """What the orchestration needs from the per-rank compute."""

    send: torch.Tensor  # Exchange.send_elems elements
    recv: torch.Tensor  # Exchange.recv_elems elements

    def start(self):
        raise NotImplementedError

    def reduce_scalars(self, read_back: bool):
        raise NotImplementedError

    def local_step(self):
        raise NotImplementedError

    def values(self) -> torch.Tensor:
        raise NotImplementedError


class HipLocalEngine(LocalEngine):
    """The product path: local CSC + fused step on the GPU through the C ABI (cugraph_amd_pagerank_mg_plan_*)."""

    def __init__(self, part: Partition, ex: Exchange, local_dst, weights, outw_local, alpha, initial_local=None):
        from . import _capi as capi
        from .pylib import GraphProperties, ResourceHandle, SGGraph, _View, assert_success

        self._capi, self._assert = capi, assert_success
        self.part = part
        dev = torch.device("cuda", torch.cuda.current_device())
        col, local_dst, outw_local = ex.col_of_edge.to(dev), local_dst.to(dev), outw_local.to(dev)
        weights = None if weights is None else weights.to(dev)
        initial_local = None if initial_local is None else initial_local.to(dev)
        self.handle = ResourceHandle()
        dtype = torch.float64 if (weights is not None and weights.dtype == torch.float64) else torch.float32
        nverts = max(ex.ncols, part.n_rows, 1)
        verts = torch.arange(nverts, dtype=torch.int32, device=dev)
        # local rows are ids [0, n_rows), columns the compact ids [0, ncols): both are vertices of the local graph
        self.graph = SGGraph(self.handle, GraphProperties(is_multigraph=True), col.to(torch.int32), local_dst.to(torch.int32),
                             weights, store_transposed=True, renumber=False, vertices_array=verts)
        self.send = torch.zeros(ex.send_elems, dtype=dtype, device=dev)
        self.recv = torch.zeros(ex.recv_elems, dtype=dtype, device=dev)
        self._outw = outw_local.to(dtype).contiguous()
        self._init = None if initial_local is None else initial_local.to(dtype).contiguous()
        self._sidx = ex.send_index.to(dev).to(torch.int32).contiguous()
        self._cpos = ex.col_pos.to(dev).contiguous()
        views = [_View(self._outw), _View(self._init), _View(self._sidx), _View(self._cpos), _View(self.send), _View(self.recv)]
        sc = (C.c_size_t * part.world)(*ex.send_counts)
        rc = (C.c_size_t * part.world)(*ex.recv_counts)
        plan, err = C.c_void_p(), C.c_void_p()
        torch.cuda.current_stream().synchronize()
        code = capi.lib().cugraph_amd_pagerank_mg_plan_create(
            self.handle.c_resource_handle_ptr, self.graph.c_graph_ptr, part.n_rows, part.nv, part.rank, part.world, views[0].ptr,
            views[1].ptr, views[2].ptr, sc, rc, views[3].ptr, views[4].ptr, views[5].ptr, float(alpha), C.byref(plan), C.byref(err))
        for v in views:
            v.free()
        assert_success(code, err, "cugraph_amd_pagerank_mg_plan_create")
        self.plan = plan
        self.dtype = dtype
        # from here on the library works on torch's current stream, the one the collectives are ordered on: the exchange and
        # the kernels that consume / produce its buffers need no host synchronisation between them
        self.shared_stream = os.environ.get("CUGRAPH_AMD_MG_OWN_STREAM") != "1"
        self.stream = _share_stream(self.handle, dev) if self.shared_stream else None

    def _call(self, name, *args):
        err = C.c_void_p()
        code = getattr(self._capi.lib(), name)(self.plan, *args, C.byref(err))
        self._assert(code, err, name)

    def start(self):
        self._call("cugraph_amd_pagerank_mg_plan_start")

    def reduce_scalars(self, read_back: bool):
        diff, dang = C.c_double(0), C.c_double(0)
        self._call("cugraph_amd_pagerank_mg_plan_reduce_scalars", 1 if read_back else 0, C.byref(diff), C.byref(dang))
        return float(diff.value), float(dang.value)

    def local_step(self):
        self._call("cugraph_amd_pagerank_mg_plan_local_step")

    def values(self):
        from .pylib import _View

        out = torch.empty(self.part.n_rows, dtype=self.dtype, device=self.send.device)
        v = _View(out) if out.numel() else None
        if v is not None:
            self._call("cugraph_amd_pagerank_mg_plan_values", v.ptr)
            v.free()
        return out

    def __del__(self):
        p = getattr(self, "plan", None)
        if p:
            self._capi.lib().cugraph_amd_pagerank_mg_plan_free(p)
            self.plan = None


# --------------------------------------------------------------------------------------- orchestration
class MGPageRank:
    """Collective: every rank of `group` constructs it with ITS slice of the edge list (external ids 0..V-1)."""

    def __init__(self, src, dst, num_vertices, weights=None, alpha=0.85, group=None, engine_factory=None, initial_guess=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        nv = int(num_vertices)
        src64, dst64 = src.to(torch.int64), dst.to(torch.int64)
        # global in-degrees (schedule) and out-weight sums (PageRank's divisor): local histogram + all-reduce
        in_deg = torch.bincount(dst64, minlength=nv)
        if weights is None:
            out_w = torch.bincount(src64, minlength=nv).to(torch.float64)
        else:
            out_w = torch.bincount(src64, weights=weights.to(torch.float64), minlength=nv)
        dist.all_reduce(in_deg, group=group)
        dist.all_reduce(out_w, group=group)
        self.part = part = Partition(in_deg, self.world, self.rank)
        pos_dst = part.pos[dst64]
        src_pos, local_dst, w = _exchange_edges(part.pos[src64], (pos_dst // self.world).to(torch.int32),
                                                pos_dst % self.world, weights, self.world, group)
        self.num_local_edges = int(src_pos.numel())
        itemsize = 8 if (w is not None and w.dtype == torch.float64) else 4
        self.ex = ex = Exchange(src_pos, self.world, self.rank, itemsize, group)
        outw_local = out_w[part.local_vertices]
        init_local = None if initial_guess is None else initial_guess[part.local_vertices]
        factory = engine_factory or HipLocalEngine
        self.engine = factory(part, ex, local_dst, w, outw_local, alpha, init_local)
        self.iterations = 0
        if getattr(self.engine, "plan", None) is not None:
            release_build_temporaries()
        with on_engine_stream(self.engine):
            self.engine.start()

    def _exchange(self):
        e, ex = self.engine, self.ex
        if e.recv.is_cuda and dist.get_backend(self.group) == "gloo":
            # test configuration (several ranks sharing one GPU): gloo moves host memory
            recv = torch.empty(e.recv.shape, dtype=e.recv.dtype)
            dist.all_to_all_single(recv, e.send.cpu(), output_split_sizes=ex.recv_splits, input_split_sizes=ex.send_splits, group=self.group)
            e.recv.copy_(recv)
        else:
            dist.all_to_all_single(e.recv, e.send, output_split_sizes=ex.recv_splits, input_split_sizes=ex.send_splits, group=self.group)
        if e.recv.is_cuda and not getattr(e, "shared_stream", False):
            torch.cuda.current_stream().synchronize()  # the library computes on its own HIP stream

    def step(self, n_iterations, epsilon=0.0):
        """Runs up to n_iterations power iterations; stops when the global L1 change drops below epsilon
        (pagerank_impl.cuh:320-326).  Returns (iterations_done, converged)."""
        with on_engine_stream(self.engine):
            done = 0
            while done < n_iterations:
                self._exchange()
                diff, _ = self.engine.reduce_scalars(epsilon > 0.0)
                if epsilon > 0.0 and self.iterations > 0 and diff < epsilon:
                    return done, True
                self.engine.local_step()
                self.iterations += 1
                done += 1
            if epsilon > 0.0:  # did the last allowed iteration converge?
                self._exchange()
                diff, _ = self.engine.reduce_scalars(True)
                return done, diff < epsilon
            return done, False

    def result(self):
        """(external vertex ids, pagerank values) of the vertices this rank owns."""
        _order_engine_stream_behind_caller(self.engine)  # the result tensor belongs to the caller's stream; the library fills it (blocking)
        return self.part.local_vertices, self.engine.values()


def pagerank(src, dst, num_vertices, weights=None, alpha=0.85, epsilon=1e-6, max_iterations=100, group=None, engine_factory=None):
    """Collective PageRank; returns (vertices, values, iterations, converged) for this rank's vertices."""
    pr = MGPageRank(src, dst, num_vertices, weights, alpha, group, engine_factory)
    done, conv = pr.step(max_iterations, epsilon)
    # detail::pagerank reports converged = iter < max_iterations (pagerank_impl.cuh:329)
    v, x = pr.result()
    return v, x, done, (conv and done < max_iterations)


# ============================================================================================================
# 2-D layout: the reference's partitioning (partition_t, cpp/include/cugraph/graph_view.hpp:63-230; partition_manager.hpp:42-51,165-178)
# behind the same orchestration, as an A/B against the 1-D sparse all-to-all above.
# ============================================================================================================
def grid_shape(world: int):
    """(R, C) with R * C = world and R the largest divisor <= sqrt(world): 1x2, 2x2, 2x4 for 2, 4, 8 ranks
    (cpp/tests/utilities/mg_utilities.cpp:48-52 picks the row size the same way)."""
    r = int(world ** 0.5)
    while world % r:
        r -= 1
    return r, world // r


class Partition2D:
    """partition_t arithmetic for P = R x C ranks, rank = c * R + r (partition_manager.hpp:42-51).

    Vertices: position p in the global descending in-degree order (ties: ascending id) -> vertex partition q = p % P, row l = p // P;
    every partition has L = ceil(V / P) rows (the last ones padded).  Rank (r, c) OWNS partition q = c * R + r = its rank.
    Edge (s, d) is STORED on the rank whose column group holds s and whose row group holds d:
        c = q_s // R,  r = q_d % R;   local column (q_s % R) * L + l_s in [0, R * L);   local row (q_d // R) * L + l_d in [0, C * L).
    Column group of rank (r, c) = ranks {c * R + r'}: their owned partitions are exactly the block's R source partitions, in local
    column order (all-gather).  Row group = ranks {c' * R + r}: member c' owns the block's destination partition c' (reduce-scatter)."""

    def __init__(self, in_degree: torch.Tensor, world: int, rank: int, shape=None):
        self.nv = int(in_degree.numel())
        self.world, self.rank = world, rank
        self.R, self.C = shape or grid_shape(world)
        assert self.R * self.C == world
        self.r, self.c = rank % self.R, rank // self.R
        order = degree_order(in_degree)
        self.order = order
        self.pos = torch.empty_like(order)
        self.pos[order] = torch.arange(self.nv, dtype=order.dtype, device=order.device)
        self.L = -(-self.nv // world)
        self.local_vertices = order[rank::world]
        self.n_rows = int(self.local_vertices.numel())  # owned (unpadded) rows
        self.col_group = [self.c * self.R + rr for rr in range(self.R)]
        self.row_group = [cc * self.R + self.r for cc in range(self.C)]

    def edge_owner(self, pos_src, pos_dst):
        P, R = self.world, self.R
        return ((pos_src % P) // R) * R + (pos_dst % P) % R

    def local_col(self, pos_src):
        return ((pos_src % self.world) % self.R) * self.L + pos_src // self.world

    def local_row(self, pos_dst):
        return ((pos_dst % self.world) // self.R) * self.L + pos_dst // self.world


class LocalEngine2D:
    """Per-rank compute of the 2-D layout.  Buffers (torch tensors): x_own [L], x_cols [R * L], y_part [C * L], y_own [L], triple [4] f64."""

    def start(self):
        raise NotImplementedError

    def set_scalars(self, gathered, read_back: bool):
        raise NotImplementedError

    def spmv(self):
        raise NotImplementedError

    def epilogue(self):
        raise NotImplementedError

    def values(self) -> torch.Tensor:
        raise NotImplementedError


class HipLocalEngine2D(LocalEngine2D):
    """The product path: local block as a CSC graph + cugraph_amd_pagerank_mg2d_plan_* (HIP)."""

    def __init__(self, part: Partition2D, local_col, local_row, weights, outw_own, alpha, initial_own=None):
        from . import _capi as capi
        from .pylib import GraphProperties, ResourceHandle, SGGraph, _View, assert_success

        self._capi, self._assert = capi, assert_success
        self.part = part
        dev = torch.device("cuda", torch.cuda.current_device())
        L, R, Cc = part.L, part.R, part.C
        weights = None if weights is None else weights.to(dev)
        dtype = torch.float64 if (weights is not None and weights.dtype == torch.float64) else torch.float32
        self.dtype = dtype
        self.handle = ResourceHandle()
        nverts = max(R * L, Cc * L, 1)
        verts = torch.arange(nverts, dtype=torch.int32, device=dev)
        self.graph = SGGraph(self.handle, GraphProperties(is_multigraph=True), local_col.to(dev).to(torch.int32), local_row.to(dev).to(torch.int32), weights,
                             store_transposed=True, renumber=False, vertices_array=verts)
        self.x_own = torch.zeros(L, dtype=dtype, device=dev)
        self.x_cols = torch.zeros(R * L, dtype=dtype, device=dev)
        self.y_part = torch.zeros(Cc * L, dtype=dtype, device=dev)
        self.y_own = torch.zeros(L, dtype=dtype, device=dev)
        self.triple = torch.zeros(4, dtype=torch.float64, device=dev)
        outw = torch.zeros(L, dtype=dtype, device=dev)
        outw[: part.n_rows] = outw_own.to(dev).to(dtype)
        init = None
        if initial_own is not None:
            init = torch.zeros(L, dtype=dtype, device=dev)
            init[: part.n_rows] = initial_own.to(dev).to(dtype)
        self._keep = (outw, init)
        views = [_View(outw), _View(init), _View(self.x_own), _View(self.x_cols), _View(self.y_part), _View(self.y_own), _View(self.triple)]
        plan, err = C.c_void_p(), C.c_void_p()
        torch.cuda.current_stream().synchronize()
        code = capi.lib().cugraph_amd_pagerank_mg2d_plan_create(
            self.handle.c_resource_handle_ptr, self.graph.c_graph_ptr, L, part.n_rows, Cc * L, R * L, part.nv, views[0].ptr, views[1].ptr, views[2].ptr, views[3].ptr,
            views[4].ptr, views[5].ptr, views[6].ptr, float(alpha), C.byref(plan), C.byref(err))
        for v in views:
            v.free()
        assert_success(code, err, "cugraph_amd_pagerank_mg2d_plan_create")
        self.plan = plan
        self.shared_stream = os.environ.get("CUGRAPH_AMD_MG_OWN_STREAM") != "1"
        self.stream = _share_stream(self.handle, dev) if self.shared_stream else None

    def _call(self, name, *args):
        err = C.c_void_p()
        code = getattr(self._capi.lib(), name)(self.plan, *args, C.byref(err))
        self._assert(code, err, name)

    def start(self):
        self._call("cugraph_amd_pagerank_mg2d_plan_start")

    def set_scalars(self, gathered, read_back):
        diff, dang = C.c_double(0), C.c_double(0)
        g = gathered.to(self.x_own.device).contiguous()
        self._call("cugraph_amd_pagerank_mg2d_plan_set_scalars", C.c_void_p(g.data_ptr()), int(g.numel() // 4), 1 if read_back else 0, C.byref(diff), C.byref(dang))
        self._last_gathered = g  # stays alive until the kernel that reads it has been ordered behind the next call
        return float(diff.value), float(dang.value)

    def spmv(self):
        self._call("cugraph_amd_pagerank_mg2d_plan_spmv")

    def epilogue(self):
        self._call("cugraph_amd_pagerank_mg2d_plan_epilogue")

    def values(self):
        from .pylib import _View

        out = torch.empty(self.part.L, dtype=self.dtype, device=self.x_own.device)
        v = _View(out)
        self._call("cugraph_amd_pagerank_mg2d_plan_values", v.ptr)
        v.free()
        return out[: self.part.n_rows]

    def __del__(self):
        p = getattr(self, "plan", None)
        if p:
            self._capi.lib().cugraph_amd_pagerank_mg2d_plan_free(p)
            self.plan = None


class MGPageRank2D:
    """Collective: every rank of the world constructs it with ITS slice of the edge list (external ids 0..V-1).  Per iteration:
    all-gather of x over the column group, local SpMV, reduce-scatter of the partial rows over the row group, epilogue on the owned
    rows, all-gather of the scalar triples (folded in rank order: deterministic)."""

    def __init__(self, src, dst, num_vertices, weights=None, alpha=0.85, group=None, engine_factory=None, initial_guess=None, shape=None):
        assert group is None, "the 2-D layout builds its row / column groups from the default process group"
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        nv = int(num_vertices)
        src64, dst64 = src.to(torch.int64), dst.to(torch.int64)
        in_deg = torch.bincount(dst64, minlength=nv)
        if weights is None:
            out_w = torch.bincount(src64, minlength=nv).to(torch.float64)
        else:
            out_w = torch.bincount(src64, weights=weights.to(torch.float64), minlength=nv)
        dist.all_reduce(in_deg)
        dist.all_reduce(out_w)
        self.part = part = Partition2D(in_deg, self.world, self.rank, shape)
        # every rank creates every group, in the same order (torch.distributed contract)
        self.col_pg = self.row_pg = None
        for c in range(part.C):
            g = dist.new_group([c * part.R + rr for rr in range(part.R)])
            if c == part.c:
                self.col_pg = g
        for r in range(part.R):
            g = dist.new_group([cc * part.R + r for cc in range(part.C)])
            if r == part.r:
                self.row_pg = g
        pos_src, pos_dst = part.pos[src64], part.pos[dst64]
        lcol, lrow, w = _exchange_edges(part.local_col(pos_src).to(torch.int32), part.local_row(pos_dst).to(torch.int32), part.edge_owner(pos_src, pos_dst),
                                        weights, self.world, None)
        self.num_local_edges = int(lcol.numel())
        outw_own = out_w[part.local_vertices]
        init_own = None if initial_guess is None else initial_guess[part.local_vertices]
        factory = engine_factory or HipLocalEngine2D
        self.engine = factory(part, lcol, lrow, w, outw_own, alpha, init_own)
        self.iterations = 0
        if getattr(self.engine, "plan", None) is not None:
            release_build_temporaries()
        with on_engine_stream(self.engine):
            self.engine.start()
        self.bytes_per_iteration = {"all_gather_in": (part.R - 1) * part.L * self.engine.x_own.element_size(),
                                    "reduce_scatter_out": (part.C - 1) * part.L * self.engine.x_own.element_size()}

    def _host(self, t):
        """gloo moves host memory (test configuration: several ranks sharing one GPU)"""
        return t.is_cuda and dist.get_backend() == "gloo"

    def _gather_x(self):
        e = self.engine
        if self._host(e.x_own):
            out = torch.empty(e.x_cols.shape, dtype=e.x_cols.dtype)
            dist.all_gather_into_tensor(out, e.x_own.cpu(), group=self.col_pg)
            e.x_cols.copy_(out)
        else:
            dist.all_gather_into_tensor(e.x_cols, e.x_own, group=self.col_pg)

    def _reduce_y(self):
        e = self.engine
        if self._host(e.y_part):
            out = torch.empty(e.y_own.shape, dtype=e.y_own.dtype)
            dist.reduce_scatter_tensor(out, e.y_part.cpu(), group=self.row_pg)
            e.y_own.copy_(out)
        else:
            dist.reduce_scatter_tensor(e.y_own, e.y_part, group=self.row_pg)

    def _gather_scalars(self, read_back):
        e = self.engine
        if self._host(e.triple):
            out = torch.empty(4 * self.world, dtype=torch.float64)
            dist.all_gather_into_tensor(out, e.triple.cpu())
        else:
            out = torch.empty(4 * self.world, dtype=torch.float64, device=e.triple.device)
            dist.all_gather_into_tensor(out, e.triple)
        if e.x_own.is_cuda and not getattr(e, "shared_stream", False):
            torch.cuda.current_stream().synchronize()
        return e.set_scalars(out, read_back)

    def _sync_for_library(self):
        e = self.engine
        if e.x_own.is_cuda and not getattr(e, "shared_stream", False):
            torch.cuda.current_stream().synchronize()  # the library computes on its own HIP stream

    def step(self, n_iterations, epsilon=0.0):
        with on_engine_stream(self.engine):
            done = 0
            while done < n_iterations:
                diff, _ = self._gather_scalars(epsilon > 0.0)
                if epsilon > 0.0 and self.iterations > 0 and diff < epsilon:
                    return done, True
                self._gather_x()
                self._sync_for_library()
                self.engine.spmv()
                self._reduce_y()
                self._sync_for_library()
                self.engine.epilogue()
                self.iterations += 1
                done += 1
            if epsilon > 0.0:
                diff, _ = self._gather_scalars(True)
                return done, diff < epsilon
            return done, False

    def result(self):
        _order_engine_stream_behind_caller(self.engine)  # the result tensor belongs to the caller's stream; the library fills it (blocking)
        return self.part.local_vertices, self.engine.values()


def pagerank_2d(src, dst, num_vertices, weights=None, alpha=0.85, epsilon=1e-6, max_iterations=100, engine_factory=None, shape=None):
    """Collective PageRank on the 2-D layout; returns (vertices, values, iterations, converged) for this rank's vertices."""
    pr = MGPageRank2D(src, dst, num_vertices, weights, alpha, None, engine_factory, None, shape)
    done, conv = pr.step(max_iterations, epsilon)
    v, x = pr.result()
    return v, x, done, (conv and done < max_iterations)


# ----------------------------------------------------------------------------------------------- bench
def bench_main(args):
    """bench.py --gpus N (N > 1): strong scaling of the SAME RMAT graph over N ranks, one per GPU."""
    from .pylib import ResourceHandle, generate_rmat_edgelist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    single = os.environ.get("CUGRAPH_AMD_MG_TEST_SINGLE_GPU") == "1"  # plumbing check: all ranks share cuda:0, gloo moves the data
    if single:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        if single:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    nv, ne = 1 << args.scale, args.edge_factor << args.scale
    h = ResourceHandle()
    if args.hot_tile is not None:
        h.set_pagerank_hot_tile(args.hot_tile)
    per = (ne + world - 1) // world
    first = rank * per
    count = max(0, min(per, ne - first))
    t0 = time.perf_counter()
    src, dst = generate_rmat_edgelist(h, args.scale, count, first_edge=first)
    if single:
        src, dst = src.cpu(), dst.cpu()  # gloo: the setup collectives run on host tensors
    layout = getattr(args, "layout", "1d")
    pr = MGPageRank2D(src, dst, nv, alpha=0.85) if layout == "2d" else MGPageRank(src, dst, nv, alpha=0.85)
    del src, dst
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    pr.step(args.warmup)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pr.step(args.steps)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if single else "cuda")
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    # HIP-event timing of this rank's two SpMV kernels over a few extra (untimed) iterations: the per-GPU roofline fraction
    eh = pr.engine.handle
    eh.kernel_timing(True)
    eh.kernel_timing_reset()
    pr.step(3)
    torch.cuda.synchronize()
    n1, ms1 = eh.kernel_timing_get("pagerank_spmv")
    n2, ms2 = eh.kernel_timing_get("pagerank_reduce")
    eh.kernel_timing(False)
    kernel_s = (ms1 / max(n1, 1) + ms2 / max(n2, 1)) / 1e3
    # phase split of an iteration (3 extra untimed iterations with a synchronisation after every phase; max over ranks):
    # exchange = the sparse all-to-all of x, scalars = folding the P message tails, local = unpack + phase 1 + phase 2 + pack
    if layout == "2d":
        # the phases of one real iteration, in order
        phases = (("scalars", lambda: pr._gather_scalars(False)), ("exchange", lambda: (pr._gather_x(), pr._sync_for_library())), ("spmv", pr.engine.spmv),
                  ("reduce_rows", lambda: (pr._reduce_y(), pr._sync_for_library())), ("epilogue", pr.engine.epilogue))
    else:
        phases = (("exchange", pr._exchange), ("reduce_scalars", lambda: pr.engine.reduce_scalars(False)), ("local_step", pr.engine.local_step))
    split = torch.zeros(len(phases), dtype=torch.float64)
    with on_engine_stream(pr.engine):
        for _ in range(3):
            for k, (_, fn) in enumerate(phases):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                split[k] += (time.perf_counter() - t1) / 3
            pr.iterations += 1
    split = split if single else split.cuda()
    dist.all_reduce(split, op=dist.ReduceOp.MAX)
    split = split.cpu().tolist()
    # outside the timed region: is the distributed vector still a probability vector, and does every rank hold one value per
    # owned vertex?  (the kernels are the single-GPU ones, checked against an explicit fp64 step by bench.py at N = 1)
    _, vals = pr.result()
    mass = vals.double().sum().reshape(1)
    rows = torch.tensor([float(vals.numel())], dtype=torch.float64, device=mass.device)
    mass = mass.cpu() if single else mass
    rows = rows.cpu() if single else rows
    dist.all_reduce(mass)
    dist.all_reduce(rows)
    check = {"mass_err": abs(float(mass.item()) - 1.0), "rows": int(rows.item()), "ok": abs(float(mass.item()) - 1.0) <= 1e-4 and int(rows.item()) == nv,
             "what": "sum of the distributed PageRank vector and number of owned rows over all ranks"}
    local_bytes = 4 * pr.num_local_edges + 16 * pr.part.n_rows + 4  # this rank's share of 4E + 16V + 4
    if layout == "2d":
        workload_tail = (f"2-D layout {pr.part.R} x {pr.part.C} (rank = c * R + r): column all-gather of x, block SpMV, row reduce-scatter of the partial rows, "
                         "scalar all-gather, over RCCL")
        exchange_info = dict(pr.bytes_per_iteration, grid=[pr.part.R, pr.part.C], rows_per_partition=pr.part.L)
    else:
        workload_tail = "1-D destination partition, degree-order round-robin, one sparse all-to-all of x per iteration over RCCL"
        exchange_info = {"columns": pr.ex.ncols, "recv_bytes_per_iteration": pr.ex.recv_elems * 4, "send_bytes_per_iteration": pr.ex.send_elems * 4}
    out = None
    if rank == 0:
        out = {
            "metric": f"pagerank_mteps_rmat{args.scale}", "value": round(ne * args.steps / dt / 1e6, 1), "unit": "MTEPS", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"PageRank power iteration, RMAT scale {args.scale} edge factor {args.edge_factor} (a,b,c)=(0.57,0.19,0.19) "
                                   "seed 0, int32 ids, fp32 ranks, alpha 0.85; " + workload_tail,
                       "vertices": nv, "edges": ne, "parallelism": f"{world} GPUs, 1 process per GPU", "layout": layout,
                       "backend": dist.get_backend(), "world_size_reported_by_backend": dist.get_world_size()},
            "iters_per_sec": round(args.steps / dt, 2), "graph_build_s": round(build_s, 3), "local_edges_rank0": pr.num_local_edges,
            "exchange_rank0": exchange_info,
            "check": check,
            "phase_split_ms": dict({name: round(split[k] * 1e3, 4) for k, (name, _) in enumerate(phases)},
                                   note="each phase bracketed by synchronisations (no overlap), max over ranks, mean of 3 iterations"),
            "roofline": {"bound": "hbm", "achieved": round(local_bytes / kernel_s / 1e9, 1) if kernel_s > 0 else None, "peak": 8000.0, "unit": "GB/s",
                         "frac": round(local_bytes / kernel_s / 1e9 / 8000.0, 4) if kernel_s > 0 else None, "traffic": None,
                         "kernel": "k_tiled_phase1 + k_tiled_phase2 on rank 0 (per-GPU share of the algorithmic bytes / its kernel time)",
                         "avg_kernel_ms": round(kernel_s * 1e3, 4), "avg_phase1_ms": round(ms1 / max(n1, 1), 4), "avg_phase2_ms": round(ms2 / max(n2, 1), 4)},
        }
    dist.barrier()
    return out
