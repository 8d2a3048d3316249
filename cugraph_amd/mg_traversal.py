"""Multi-GPU BFS / SSSP: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).

Replaces the multi-GPU path of cugraph_bfs / cugraph_sssp ("one all-to-all per BFS/SSSP level";
cpp/src/traversal/bfs_impl.cuh:133-870 and sssp_impl.cuh:169-566 with multi_gpu = true, whose frontier expansion ends in a
shuffle of (dst, payload) pairs to the dst owner, prims/transform_reduce_if_v_frontier_outgoing_e_by_dst.cuh:617-1127).

Partitioning: vertices ordered by descending global OUT-degree (ties: ascending id) are dealt round-robin to the P ranks
(position p -> owner p % P, local row p // P): every rank gets the same mix of hub and tail rows.  1-D by source: the
owner of a vertex holds all its out-edges, so a frontier vertex is expanded on exactly one rank.  A vertex is named by its
compact global id g = owner * L + row (L = rows per rank, padded to a multiple of 64): the owner is g // L and the
concatenation of the ranks' L-bit bitmaps is the global bitmap.

Per level / relaxation round (collective):
    expand      local frontier -> candidates, reduced at the sender per destination, bucketed by owner
    all-to-all  counts, then the tuples (one message per peer: all xGMI links of a GPU carry traffic at once)
    apply       the owner folds the candidates into its rows (order-independent) and builds its next frontier
    BFS only    all-gather of the L-bit new-frontier bitmaps -> every rank's visited bitmap stays exact, so a vertex is
                sent at most once per rank in the whole search
    all-reduce  of the next-frontier sizes: the search ends when the global frontier is empty.
SSSP is a frontier Bellman-Ford; its fixed point equals Dijkstra's, so distances are bit-identical to the single-GPU path.
Parents: minimum external id among the valid parents -- independent of P.

The per-rank compute sits behind `TraversalEngine`; the one engine of this package is `HipTraversalEngine` (C ABI, HIP
kernels of csrc/traversal_mg.hip) -- there is no CPU engine here.  The tests plug a numpy engine of their own
(tests/numpy_traversal_engine.py) into the same orchestration to exercise partitioning and collectives under gloo on CPU.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from .mg import Partition, _a2a, _count_owners, inclusive_counts, release_build_temporaries, stable_argsort

INT32_MAX = 2**31 - 1
FLT_MAX = float(np.finfo(np.float32).max)


class TraversalEngine:
    """What the orchestration needs from the per-rank compute.  Tuples are int32 rows: BFS (row, parent), SSSP (row,
    distance bits, parent + 1)."""

    tuple_words: int

    def reset(self, source_rows: torch.Tensor, cutoff: float, with_pred: bool) -> int:
        raise NotImplementedError

    def expand(self):
        """-> (send tensor [n, tuple_words] grouped by owner, counts per owner)"""
        raise NotImplementedError

    def apply(self, recv: torch.Tensor, level: int) -> int:
        raise NotImplementedError

    def frontier_bits(self) -> torch.Tensor:  # BFS: int32 words, L / 32
        raise NotImplementedError

    def merge_visited(self, gathered: torch.Tensor):
        raise NotImplementedError

    # BFS, bottom-up levels (optional: an engine without them returns False and the search stays top-down)
    def enable_bottom_up(self, in_offsets: torch.Tensor, in_indices: torch.Tensor, ext_of_g: torch.Tensor) -> bool:
        return False

    def bottom_up(self, front_bits: torch.Tensor, level: int) -> int:
        """every unvisited local row scans its in-neighbours against the all-gathered frontier bitmap -> discoveries"""
        raise NotImplementedError

    def degree_sums(self):
        """(out-degree sum, in-degree sum) of the vertices the last apply / bottom_up discovered on this rank"""
        return 0, 0

    def results(self):
        raise NotImplementedError


class HipTraversalEngine(TraversalEngine):
    """The product path: the kernels of csrc/traversal_mg.hip through the C ABI (cugraph_amd_traversal_mg_plan_*)."""

    def __init__(self, mode, offsets, indices, weights, n_rows, L, rank, world, row_vertex):
        from . import _capi as capi
        from .pylib import ResourceHandle, assert_success

        self._capi, self._assert = capi, assert_success
        dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        self.mode, self.n_rows, self.L, self.world = mode, n_rows, L, world
        self.tuple_words = 2 if mode == 0 else 3
        self.handle = ResourceHandle()
        self._off = offsets.to(dev).to(torch.int32).contiguous()
        self._idx = indices.to(dev).to(torch.int32).contiguous()
        if self._idx.numel() == 0:
            self._idx = torch.zeros(1, dtype=torch.int32, device=dev)
        self._w = None if weights is None else weights.to(dev).to(torch.float32).contiguous()
        self._rv = row_vertex.to(dev).to(torch.int32).contiguous()
        if self._rv.numel() == 0:
            self._rv = torch.zeros(1, dtype=torch.int32, device=dev)
        n_edges = int(indices.numel())
        self.capacity = max(min(L * world, max(n_edges, 1)), 1)
        self.send = torch.zeros((self.capacity, self.tuple_words), dtype=torch.int32, device=dev)
        plan, err = C.c_void_p(), C.c_void_p()
        torch.cuda.current_stream().synchronize()
        code = capi.lib().cugraph_amd_traversal_mg_plan_create(
            self.handle.c_resource_handle_ptr, self._off.data_ptr(), self._idx.data_ptr(), (self._w.data_ptr() if self._w is not None else None),
            n_rows, n_edges, L, rank, world, self._rv.data_ptr(), mode, self.send.data_ptr(), self.capacity, C.byref(plan), C.byref(err))
        assert_success(code, err, "cugraph_amd_traversal_mg_plan_create")
        self.plan = plan

    def _call(self, name, *args):
        err = C.c_void_p()
        code = getattr(self._capi.lib(), name)(self.plan, *args, C.byref(err))
        self._assert(code, err, name)

    def reset(self, source_rows, cutoff, with_pred):
        self.with_pred = with_pred
        src = source_rows.to(self.device).to(torch.int32).contiguous()
        torch.cuda.current_stream().synchronize()
        self._call("cugraph_amd_traversal_mg_plan_reset", (src.data_ptr() if src.numel() else None), int(src.numel()), float(cutoff), 1 if with_pred else 0)

    def expand(self):
        counts = (C.c_size_t * self.world)()
        self._call("cugraph_amd_traversal_mg_plan_expand", counts)
        counts = [int(c) for c in counts]
        return self.send[: sum(counts)], counts

    def apply(self, recv, level):
        n_next = C.c_size_t(0)
        recv = recv.contiguous()
        torch.cuda.current_stream().synchronize()  # the library computes on its own HIP stream
        self._call("cugraph_amd_traversal_mg_plan_apply", (recv.data_ptr() if recv.numel() else None), int(recv.shape[0]), int(level), C.byref(n_next))
        return int(n_next.value)

    def frontier_bits(self):
        ptr = C.c_void_p()
        self._call("cugraph_amd_traversal_mg_plan_frontier_bits", C.byref(ptr))
        out = torch.empty(self.L // 32, dtype=torch.int32, device=self.device)
        _memcpy_d2d(out, ptr.value, self.L // 8)
        return out

    def merge_visited(self, gathered):
        g = gathered.to(self.device).contiguous()
        torch.cuda.current_stream().synchronize()
        self._call("cugraph_amd_traversal_mg_plan_merge_visited", g.data_ptr())

    def enable_bottom_up(self, in_offsets, in_indices, ext_of_g):
        dev = self.device
        self._in_off = in_offsets.to(dev).to(torch.int32).contiguous()
        # (the kernel reads 16-byte chunks of neighbour ids: over-allocate)
        self._in_idx = torch.cat([in_indices.to(dev).to(torch.int32), torch.zeros(64, dtype=torch.int32, device=dev)]).contiguous()
        self._ext_of_g = ext_of_g.to(dev).to(torch.int32).contiguous()
        torch.cuda.current_stream().synchronize()
        self._call("cugraph_amd_traversal_mg_plan_set_bottom_up", self._in_off.data_ptr(), self._in_idx.data_ptr(), self._ext_of_g.data_ptr())
        return True

    def bottom_up(self, front_bits, level):
        n = C.c_size_t(0)
        f = front_bits.to(self.device).contiguous()
        torch.cuda.current_stream().synchronize()
        self._call("cugraph_amd_traversal_mg_plan_bottom_up", f.data_ptr(), int(level), C.byref(n))
        return int(n.value)

    def degree_sums(self):
        o, i = C.c_ulonglong(0), C.c_ulonglong(0)
        self._call("cugraph_amd_traversal_mg_plan_last_degree_sums", C.byref(o), C.byref(i))
        return int(o.value), int(i.value)

    def results(self):
        n = max(self.n_rows, 1)
        d = torch.empty(n, dtype=(torch.int32 if self.mode == 0 else torch.float32), device=self.device)
        p = torch.empty(n, dtype=torch.int32, device=self.device) if self.with_pred else None
        torch.cuda.current_stream().synchronize()
        self._call("cugraph_amd_traversal_mg_plan_results", d.data_ptr(), (p.data_ptr() if p is not None else None))
        return d[: self.n_rows], (p[: self.n_rows] if p is not None else None)

    def __del__(self):
        p = getattr(self, "plan", None)
        if p:
            self._capi.lib().cugraph_amd_traversal_mg_plan_free(p)
            self.plan = None


def _memcpy_d2d(dst: torch.Tensor, src_ptr: int, nbytes: int):
    """Copies nbytes of library-owned device memory into a torch tensor (plumbing: torch owns the collective buffers)."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    rc = hip.hipMemcpy(dst.data_ptr(), src_ptr, nbytes, 3)  # hipMemcpyDeviceToDevice
    if rc != 0:
        raise RuntimeError(f"hipMemcpy failed with {rc}")


# --------------------------------------------------------------------------------------- orchestration
class MGTraversal:
    """Collective: every rank of `group` constructs it with ITS slice of the edge list (external ids 0..V-1)."""

    def __init__(self, src, dst, num_vertices, weights=None, mode="bfs", group=None, engine_factory=None):
        assert mode in ("bfs", "sssp")
        self.group = group
        self.world = world = dist.get_world_size(group)
        self.rank = rank = dist.get_rank(group)
        self.mode = 0 if mode == "bfs" else 1
        if self.mode == 1:
            assert weights is not None, "SSSP needs weights"
        nv = int(num_vertices)
        self.nv = nv
        src64, dst64 = src.to(torch.int64), dst.to(torch.int64)
        out_deg = torch.bincount(src64, minlength=nv)
        dist.all_reduce(out_deg, group=group)
        self.part = part = Partition(out_deg, world, rank)
        self.L = L = ((nv + world - 1) // world + 63) // 64 * 64
        pos_s, pos_d = part.pos[src64], part.pos[dst64]
        g_dst = (pos_d % world) * L + pos_d // world                 # compact global id of the destination
        owner = pos_s % world
        order = stable_argsort(owner, world - 1)  # the library's radix sort on device tensors (mg.py)
        send_counts = _count_owners(owner, world)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=group)
        sc, rc = send_counts.tolist(), recv_counts.tolist()
        rows = _a2a((pos_s // world)[order].contiguous(), sc, rc, group)
        cols = _a2a(g_dst[order].contiguous(), sc, rc, group)
        w = None if weights is None else _a2a(weights.to(torch.float32)[order].contiguous(), sc, rc, group)
        # local CSR (rows ascending, destinations ascending inside a row)
        n_rows = part.n_rows
        key = rows * (L * world) + cols
        o2 = stable_argsort(key, n_rows * L * world)
        rows, cols = rows[o2], cols[o2]
        w = None if w is None else w[o2].contiguous()
        offsets = torch.zeros(n_rows + 1, dtype=torch.int64, device=rows.device)
        if rows.numel():
            offsets[1:] = inclusive_counts(rows, n_rows)
        self.num_local_edges = int(cols.numel())
        assert self.num_local_edges < 2**31 and L * world < 2**31
        factory = engine_factory or HipTraversalEngine
        self.engine = factory(self.mode, offsets.to(torch.int32), cols.to(torch.int32), w, n_rows, L, rank, world, part.local_vertices.to(torch.int32))
        self.levels = 0
        self.bottom_up_levels = 0
        # BFS: bottom-up levels need the in-edges of the owned vertices -- a second copy of the edge list, routed by the owner of the
        # DESTINATION, neighbours as compact global ids in ascending order of their external id (first frontier member of a row = the
        # minimum-external-id parent, the rule of the top-down levels) -- and the external id of every compact global id
        self.can_bottom_up = False
        if self.mode == 0 and os.environ.get("CUGRAPH_AMD_MG_BFS_BOTTOM_UP", "1") != "0":
            owner_d = pos_d % world
            od = stable_argsort(owner_d, world - 1)
            sc_d = _count_owners(owner_d, world)
            rc_d = torch.empty_like(sc_d)
            dist.all_to_all_single(rc_d, sc_d, group=group)
            sc_d, rc_d = sc_d.tolist(), rc_d.tolist()
            in_rows = _a2a((pos_d // world)[od].contiguous(), sc_d, rc_d, group)
            in_g = _a2a(((pos_s % world) * L + pos_s // world)[od].contiguous(), sc_d, rc_d, group)
            in_ext = _a2a(src64[od].contiguous(), sc_d, rc_d, group)
            o3 = stable_argsort(in_rows * nv + in_ext, max(n_rows, 1) * nv)
            in_rows, in_g = in_rows[o3], in_g[o3]
            in_off = torch.zeros(n_rows + 1, dtype=torch.int64, device=in_rows.device)
            if in_rows.numel():
                in_off[1:] = inclusive_counts(in_rows, n_rows)
            mine = torch.full((L,), -1, dtype=torch.int32, device=part.local_vertices.device)
            mine[:n_rows] = part.local_vertices.to(torch.int32)
            mine_c = self._to_comm(mine)
            ext_of_g = torch.empty(world * L, dtype=torch.int32, device=mine_c.device)
            dist.all_gather_into_tensor(ext_of_g, mine_c, group=group)
            self.can_bottom_up = bool(self.engine.enable_bottom_up(in_off.to(torch.int32), in_g.to(torch.int32), ext_of_g))
            ne_t = torch.tensor([int(src64.numel())], dtype=torch.int64, device=mine_c.device)
            dist.all_reduce(ne_t, group=group)
            self.ne_global = int(ne_t.item())
            self._out_deg = out_deg
        if getattr(self.engine, "plan", None) is not None:
            release_build_temporaries()  # (the library's cache of construction temporaries: torch / RCCL share the device)

    # -- collectives on device tensors (nccl) or through the host (gloo moves host memory)
    def _to_comm(self, t):
        return t.cpu() if (t.is_cuda and dist.get_backend(self.group) == "gloo") else t

    def _exchange(self, send, counts):
        """All-to-all-v of the candidate tuples: counts first, then one message per peer."""
        dev, tw = send.device, self.engine.tuple_words
        host = dist.get_backend(self.group) == "gloo"
        sc = torch.tensor(counts, dtype=torch.int64, device=("cpu" if host else dev))
        rc = torch.empty_like(sc)
        dist.all_to_all_single(rc, sc, group=self.group)
        rc = rc.tolist()
        s = self._to_comm(send.reshape(-1))
        if s.is_cuda:
            torch.cuda.current_stream().synchronize()
        r = torch.empty(sum(rc) * tw, dtype=torch.int32, device=s.device)
        dist.all_to_all_single(r, s, output_split_sizes=[c * tw for c in rc], input_split_sizes=[c * tw for c in counts], group=self.group)
        return r.reshape(-1, tw).to(dev)

    def _share_frontier_bits(self, stats=(0, 0, 0)):
        """All-gather of the ranks' new-frontier bits; three int64 per rank (discoveries, their out- and in-degree sums) ride behind the
        bits, so the level needs no separate all-reduce.  Returns the three sums over all ranks."""
        bits = self._to_comm(self.engine.frontier_bits())
        W = bits.numel()
        tail = torch.tensor([int(x) for x in stats], dtype=torch.int64).view(torch.int32).to(bits.device)
        msg = torch.cat([bits, tail])
        out = torch.empty(self.world * msg.numel(), dtype=msg.dtype, device=msg.device)
        dist.all_gather_into_tensor(out, msg, group=self.group)
        out = out.view(self.world, W + 6)
        gathered = out[:, :W].contiguous().view(-1)
        self.engine.merge_visited(gathered)
        self._front_bits = gathered  # the frontier of the next level (what a bottom-up level tests against)
        return out[:, W:].contiguous().view(torch.int64).view(self.world, 3).sum(0).tolist()

    def run(self, sources, cutoff=FLT_MAX, compute_predecessors=True, depth_limit=None):
        """sources: external vertex ids (the same list on every rank).  Returns (vertices, distances, predecessors) of the
        vertices this rank owns.  BFS: depth_limit as cugraph_bfs; SSSP: cutoff as cugraph_sssp."""
        e, part, world = self.engine, self.part, self.world
        sources = torch.as_tensor(sources, dtype=torch.int64).reshape(-1)
        if sources.numel() and (int(sources.min()) < 0 or int(sources.max()) >= self.nv):
            raise ValueError("Found invalid vertex in the input sources")  # bfs.cpp:106-119
        pos = part.pos[sources.to(part.pos.device)].cpu()
        mine = pos[pos % world == self.rank] // world
        e.reset(mine.to(torch.int32), cutoff, compute_predecessors)
        if self.mode == 0:
            self._share_frontier_bits()
        # Beamer's direction heuristic on GLOBAL sums (every rank takes the same decision): bottom-up when the frontier's out-edges exceed
        # 1 / alpha of the unvisited vertices' in-edges, top-down again when the frontier has shrunk below V / beta (the single-GPU
        # driver's constants: traversal.hip run_bfs)
        bu_ok = self.mode == 0 and self.can_bottom_up
        alpha, beta = float(os.environ.get("CUGRAPH_AMD_BFS_ALPHA", "60")), float(os.environ.get("CUGRAPH_AMD_BFS_BETA", "24"))
        force = os.environ.get("CUGRAPH_AMD_MG_BFS", "")  # "bottomup" / "topdown": pin the direction (tests)
        n_front = int(sources.numel())
        frontier_out = int(self._out_deg[sources.to(self._out_deg.device)].sum()) if bu_ok else 0
        unvisited_in = self.ne_global if bu_ok else 0
        bottom_up = False
        comm_dev = "cpu" if dist.get_backend(self.group) == "gloo" else e.device
        level = 0
        self.bottom_up_levels = 0
        while True:
            level += 1
            if self.mode == 0 and depth_limit is not None and level > depth_limit:
                break
            if bu_ok:
                if not bottom_up:
                    bottom_up = frontier_out > unvisited_in / alpha and n_front > 1024
                else:
                    bottom_up = not (n_front < self.nv / beta)
                if force:
                    bottom_up = force == "bottomup"
            if bottom_up:
                n_next = e.bottom_up(self._front_bits, level)
                self.bottom_up_levels += 1
            else:
                send, counts = e.expand()
                recv = self._exchange(send, counts)
                n_next = e.apply(recv, level)
            o_sum, i_sum = e.degree_sums() if bu_ok else (0, 0)
            if self.mode == 0:
                n_front, frontier_out, i_tot = self._share_frontier_bits((n_next, o_sum, i_sum))
                unvisited_in = max(0, unvisited_in - int(i_tot))
            else:
                tot = torch.tensor([n_next], dtype=torch.int64, device=comm_dev)
                dist.all_reduce(tot, group=self.group)
                n_front = int(tot.item())
            if n_front == 0:
                break
        self.levels = level
        d, p = e.results()
        return part.local_vertices, d, p


def bfs(src, dst, num_vertices, sources, depth_limit=None, compute_predecessors=True, group=None, engine_factory=None):
    """Collective BFS; returns (vertices, distances, predecessors) for this rank's vertices."""
    t = MGTraversal(src, dst, num_vertices, None, "bfs", group, engine_factory)
    return t.run(sources, compute_predecessors=compute_predecessors, depth_limit=depth_limit)


def sssp(src, dst, weights, num_vertices, source, cutoff=FLT_MAX, compute_predecessors=True, group=None, engine_factory=None):
    """Collective SSSP (float weights); returns (vertices, distances, predecessors) for this rank's vertices."""
    t = MGTraversal(src, dst, num_vertices, weights, "sssp", group, engine_factory)
    return t.run([source], cutoff=cutoff, compute_predecessors=compute_predecessors)
