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

import ctypes as C
import os
import time

import torch
import torch.distributed as dist

TAIL_BYTES = 32  # (L1 change, dangling mass, max |x|) as 3 doubles + padding, behind every message


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
    order = torch.argsort(owner_dst, stable=True)
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
        need, self.col_of_edge = torch.unique(src_pos, sorted=True, return_inverse=True)
        self.ncols = int(need.numel())
        owner = need % world
        order = torch.argsort(owner, stable=True)            # requests grouped by owner, ascending position inside
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
        if self.shared_stream:
            self.handle.set_stream(torch.cuda.current_stream().cuda_stream)

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
    split = torch.zeros(3, dtype=torch.float64)
    for _ in range(3):
        for k, fn in enumerate((pr._exchange, lambda: pr.engine.reduce_scalars(False), pr.engine.local_step)):
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
    out = None
    if rank == 0:
        out = {
            "metric": f"pagerank_mteps_rmat{args.scale}", "value": round(ne * args.steps / dt / 1e6, 1), "unit": "MTEPS", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"PageRank power iteration, RMAT scale {args.scale} edge factor {args.edge_factor} (a,b,c)=(0.57,0.19,0.19) "
                                   "seed 0, int32 ids, fp32 ranks, alpha 0.85; 1-D destination partition, degree-order round-robin, "
                                   "one sparse all-to-all of x per iteration over RCCL",
                       "vertices": nv, "edges": ne, "parallelism": f"{world} GPUs, 1 process per GPU"},
            "iters_per_sec": round(args.steps / dt, 2), "graph_build_s": round(build_s, 3), "local_edges_rank0": pr.num_local_edges,
            "exchange_rank0": {"columns": pr.ex.ncols, "recv_bytes_per_iteration": pr.ex.recv_elems * 4, "send_bytes_per_iteration": pr.ex.send_elems * 4},
            "check": check,
            "phase_split_ms": {"exchange": round(split[0] * 1e3, 4), "reduce_scalars": round(split[1] * 1e3, 4), "local_step": round(split[2] * 1e3, 4),
                               "note": "each phase bracketed by synchronisations (no overlap), max over ranks, mean of 3 iterations"},
            "roofline": {"bound": "hbm", "achieved": round(local_bytes / kernel_s / 1e9, 1) if kernel_s > 0 else None, "peak": 8000.0, "unit": "GB/s",
                         "frac": round(local_bytes / kernel_s / 1e9 / 8000.0, 4) if kernel_s > 0 else None, "traffic": None,
                         "kernel": "k_tiled_phase1 + k_tiled_phase2 on rank 0 (per-GPU share of the algorithmic bytes / its kernel time)",
                         "avg_kernel_ms": round(kernel_s * 1e3, 4), "avg_phase1_ms": round(ms1 / max(n1, 1), 4), "avg_phase2_ms": round(ms2 / max(n2, 1), 4)},
        }
    dist.barrier()
    return out
