"""Multi-GPU PageRank: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).

Partitioning (re-designed for a full-mesh xGMI node instead of translated):
  * vertices are ordered by descending GLOBAL in-degree (ties: ascending id) and dealt round-robin to the P ranks:
    position p -> owner p % P, local row p // P.  Every rank gets the same mix of hub and tail rows (balanced edge
    counts) and its local rows are already degree-sorted, which is what the edge-balanced SpMV kernel wants;
  * 1-D by destination: the owner of a destination holds ALL its in-edges, so the pull-SpMV needs no partial-sum
    reduction over ranks (the reference's 2-D scheme needs a row broadcast AND a column reduce per iteration,
    prims/update_edge_src_dst_property.cuh:550-579 + prims/detail/per_v_transform_reduce_e.cuh:3390-3406);
  * per iteration ONE collective: all-gather of x = pr / out_w (chunk of V/P values per rank) with the two scalars of
    the iteration (partial L1 change, partial dangling mass, max |x|) riding in the last 32 bytes of every chunk -- no scalar
    all-reduce, every rank adds the P partials in rank order (deterministic).
Column ids stored in the local CSC are the global degree-order positions, so the hottest sources are ids [0, K)
and the LDS hot tile of the SpMV kernel keeps working across ranks.

The local compute sits behind `LocalEngine`; the product engine is `HipLocalEngine` (C ABI, HIP).  Tests plug a
CPU engine built on the oracle into the same orchestration to exercise partitioning + collectives under gloo.
"""
from __future__ import annotations

import ctypes as C
import os
import time

import torch
import torch.distributed as dist


# ------------------------------------------------------------------------------------------ partition
class Partition:
    """Global degree-order numbering dealt round-robin over the ranks."""

    def __init__(self, in_degree: torch.Tensor, world: int, rank: int):
        self.nv = int(in_degree.numel())
        self.world, self.rank = world, rank
        # stable descending sort: ties keep ascending vertex id
        _, order = torch.sort(in_degree, descending=True, stable=True)
        self.order = order                                   # position -> vertex
        self.pos = torch.empty_like(order)
        self.pos[order] = torch.arange(self.nv, dtype=order.dtype, device=order.device)  # vertex -> position (= column id)
        self.local_vertices = order[rank::world]             # external ids of the rows this rank owns, local order
        self.n_rows = int(self.local_vertices.numel())
        lmax = (self.nv + world - 1) // world
        self.chunk = (lmax + 8 + 3) // 4 * 4                 # local rows + 32 B of scalars (3 doubles + pad), 16-byte multiple
        self.ncols = self.chunk * world


def _exchange_edges(col_src, local_dst, owner_dst, weights, world, group):
    """Routes every edge to the owner of its destination (all-to-all-v), like shuffle_ext_edges in the reference
    (cpp/src/c_api/graph_mg.cpp:140) but keyed on the degree-order owner instead of a hash."""
    order = torch.argsort(owner_dst, stable=True)
    col_src, local_dst = col_src[order].contiguous(), local_dst[order].contiguous()
    if weights is not None:
        weights = weights[order].contiguous()
    send_counts = torch.bincount(owner_dst, minlength=world).to(torch.int64)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)
    sc, rc = send_counts.tolist(), recv_counts.tolist()
    n_recv = int(sum(rc))

    def a2a(t):
        out = torch.empty(n_recv, dtype=t.dtype, device=t.device)
        dist.all_to_all_single(out, t, output_split_sizes=rc, input_split_sizes=sc, group=group)
        return out

    return a2a(col_src), a2a(local_dst), (a2a(weights) if weights is not None else None)


# --------------------------------------------------------------------------------------------- engines
class LocalEngine:
    """What the orchestration needs from the per-rank compute."""

    send: torch.Tensor  # chunk elements
    recv: torch.Tensor  # world * chunk elements

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

    def __init__(self, part: Partition, col_src, local_dst, weights, outw_local, alpha, initial_local=None):
        from . import _capi as capi
        from .pylib import GraphProperties, ResourceHandle, SGGraph, _View, assert_success

        self._capi, self._assert = capi, assert_success
        self.part = part
        dev = torch.device("cuda", torch.cuda.current_device())
        col_src, local_dst, outw_local = col_src.to(dev), local_dst.to(dev), outw_local.to(dev)
        weights = None if weights is None else weights.to(dev)
        initial_local = None if initial_local is None else initial_local.to(dev)
        self.handle = ResourceHandle()
        dtype = torch.float64 if (weights is not None and weights.dtype == torch.float64) else torch.float32
        cols = torch.arange(part.ncols, dtype=torch.int32, device=dev)
        # local rows are ids [0, n_rows); ncols >= n_rows vertices so that every column id is a vertex of the local graph
        self.graph = SGGraph(self.handle, GraphProperties(is_multigraph=True), col_src.to(torch.int32), local_dst.to(torch.int32),
                             weights, store_transposed=True, renumber=False, vertices_array=cols)
        chunk = part.chunk if dtype == torch.float32 else part.chunk  # fp64: same element count (32-byte multiple)
        self.send = torch.zeros(chunk, dtype=dtype, device=dev)
        self.recv = torch.zeros(chunk * part.world, dtype=dtype, device=dev)
        self._outw = outw_local.to(dtype).contiguous()
        self._init = None if initial_local is None else initial_local.to(dtype).contiguous()
        views = [_View(self._outw), _View(self._init), _View(self.send), _View(self.recv)]
        plan, err = C.c_void_p(), C.c_void_p()
        torch.cuda.current_stream().synchronize()
        code = capi.lib().cugraph_amd_pagerank_mg_plan_create(
            self.handle.c_resource_handle_ptr, self.graph.c_graph_ptr, part.n_rows, part.nv, part.rank, part.world, chunk, views[0].ptr,
            views[1].ptr, views[2].ptr, views[3].ptr, float(alpha), C.byref(plan), C.byref(err))
        for v in views:
            v.free()
        assert_success(code, err, "cugraph_amd_pagerank_mg_plan_create")
        self.plan = plan
        self.dtype = dtype

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
        dev = src.device
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
        col_src, local_dst, w = _exchange_edges(part.pos[src64].to(torch.int32), (pos_dst // self.world).to(torch.int32),
                                                pos_dst % self.world, weights, self.world, group)
        self.num_local_edges = int(col_src.numel())
        outw_local = out_w[part.local_vertices]
        init_local = None if initial_guess is None else initial_guess[part.local_vertices]
        factory = engine_factory or HipLocalEngine
        self.engine = factory(part, col_src, local_dst, w, outw_local, alpha, init_local)
        self.iterations = 0
        self.engine.start()

    def _gather(self):
        e = self.engine
        if e.recv.is_cuda and dist.get_backend(self.group) == "gloo":
            # test configuration (several ranks sharing one GPU): gloo moves host memory
            recv = torch.empty(e.recv.shape, dtype=e.recv.dtype)
            dist.all_gather_into_tensor(recv, e.send.cpu(), group=self.group)
            e.recv.copy_(recv)
        else:
            dist.all_gather_into_tensor(e.recv, e.send, group=self.group)
        if e.recv.is_cuda:
            torch.cuda.current_stream().synchronize()  # the library computes on its own HIP stream

    def step(self, n_iterations, epsilon=0.0):
        """Runs up to n_iterations power iterations; stops when the global L1 change drops below epsilon
        (pagerank_impl.cuh:320-326).  Returns (iterations_done, converged)."""
        done = 0
        while done < n_iterations:
            self._gather()
            diff, _ = self.engine.reduce_scalars(epsilon > 0.0)
            if epsilon > 0.0 and self.iterations > 0 and diff < epsilon:
                return done, True
            self.engine.local_step()
            self.iterations += 1
            done += 1
        if epsilon > 0.0:  # did the last allowed iteration converge?
            self._gather()
            diff, _ = self.engine.reduce_scalars(True)
            return done, diff < epsilon
        return done, False

    def result(self):
        """(external vertex ids, pagerank values) of the vertices this rank owns."""
        return self.part.local_vertices, self.engine.values()


def pagerank(src, dst, num_vertices, weights=None, alpha=0.85, epsilon=1e-6, max_iterations=100, group=None, engine_factory=None):
    """Collective PageRank; returns (vertices, values, iterations, converged) for this rank's vertices."""
    pr = MGPageRank(src, dst, num_vertices, weights, alpha, group, engine_factory)
    done, conv = pr.step(max_iterations, epsilon)
    # detail::pagerank reports converged = iter < max_iterations (pagerank_impl.cuh:329)
    v, x = pr.result()
    return v, x, done, (conv and done < max_iterations)


# ----------------------------------------------------------------------------------------------- bench
def bench_main(args):
    """bench.py --gpus N (N > 1): strong scaling of the SAME RMAT graph over N ranks, one per GPU."""
    from .pylib import ResourceHandle, generate_rmat_edgelist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
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
    pr = MGPageRank(src, dst, nv, alpha=0.85)
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
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    out = None
    if rank == 0:
        out = {
            "metric": f"pagerank_mteps_rmat{args.scale}", "value": round(ne * args.steps / dt / 1e6, 1), "unit": "MTEPS", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"PageRank power iteration, RMAT scale {args.scale} edge factor {args.edge_factor} (a,b,c)=(0.57,0.19,0.19) "
                                   "seed 0, int32 ids, fp32 ranks, alpha 0.85; 1-D destination partition, degree-order round-robin, "
                                   "one all-gather of x per iteration over RCCL",
                       "vertices": nv, "edges": ne, "parallelism": f"{world} GPUs, 1 process per GPU"},
            "iters_per_sec": round(args.steps / dt, 2), "graph_build_s": round(build_s, 3), "local_edges_rank0": pr.num_local_edges,
        }
    dist.barrier()
    return out
