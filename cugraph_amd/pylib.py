"""Host-side mirror of the reference's pylibcugraph interface for the PageRank / BFS / SSSP path.

Same names, argument order, return tuples and exceptions as
  python/pylibcugraph/pylibcugraph/{resource_handle,graph_properties,graphs,pagerank,personalized_pagerank,
  bfs,sssp,has_vertex}.pyx and utils.pyx:assert_success,
so the reference's pylibcugraph tests read unchanged against this module.  Device arrays are torch tensors
on the HIP device (the reference uses cupy); anything exposing __cuda_array_interface__ is accepted too.
Everything computes through the C ABI (cugraph_amd/_capi.py); PyTorch is used for device memory only.
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np
import torch

from . import _capi as capi

INT_MAX = 2147483647

_TORCH_TO_C = {
    torch.int8: capi.INT8, torch.int16: capi.INT16, torch.int32: capi.INT32, torch.int64: capi.INT64,
    torch.uint8: capi.UINT8, torch.float32: capi.FLOAT32, torch.float64: capi.FLOAT64, torch.bool: capi.BOOL,
}
_C_TO_TORCH = {
    capi.INT8: torch.int8, capi.INT16: torch.int16, capi.INT32: torch.int32, capi.INT64: torch.int64,
    capi.UINT8: torch.uint8, capi.FLOAT32: torch.float32, capi.FLOAT64: torch.float64, capi.BOOL: torch.bool,
    capi.SIZE_T: torch.int64,
}
_NP_TO_C = {"int8": capi.INT8, "int16": capi.INT16, "int32": capi.INT32, "int64": capi.INT64, "uint8": capi.UINT8,
            "uint16": capi.UINT16, "uint32": capi.UINT32, "uint64": capi.UINT64, "float32": capi.FLOAT32,
            "float64": capi.FLOAT64, "bool": capi.BOOL}



def _lib_at_exit():
    """the library for a finalizer: None once the interpreter has torn this module's globals down (objects that outlive it are released with the process)"""
    try:
        return capi.lib()
    except Exception:  # noqa: BLE001
        return None

class FailedToConvergeError(Exception):
    """Raised when an iterative algorithm does not converge (pylibcugraph/exceptions.py:9)."""


def assert_success(code, err, api_name):
    """utils.pyx:36-83: error code -> Python exception of the same class as the reference raises."""
    if code == capi.CUGRAPH_SUCCESS:
        return
    l = capi.lib()
    msg = l.cugraph_error_message(err)
    msg = msg.decode() if isinstance(msg, bytes) else str(msg)
    l.cugraph_error_free(err)
    name = capi.ERROR_NAMES[code] if 0 <= code < len(capi.ERROR_NAMES) else "unknown error code"
    text = f"non-success value returned from {api_name}: {name} {msg}"
    if code in (capi.CUGRAPH_INVALID_HANDLE, capi.CUGRAPH_INVALID_INPUT, capi.CUGRAPH_UNSUPPORTED_TYPE_COMBINATION):
        raise ValueError(text)
    if code == capi.CUGRAPH_ALLOC_ERROR:
        raise MemoryError(text)
    if code == capi.CUGRAPH_NOT_IMPLEMENTED:
        raise NotImplementedError(text)
    raise RuntimeError(text)


def _describe(obj):
    """(device pointer, length, C type id) of a torch tensor or a __cuda_array_interface__ object."""
    if isinstance(obj, torch.Tensor):
        if not obj.is_cuda:
            raise TypeError("expected a device (cuda/HIP) tensor")
        if not obj.is_contiguous() or obj.dim() != 1:
            raise TypeError("expected a contiguous 1-D tensor")
        return obj.data_ptr(), obj.numel(), _TORCH_TO_C[obj.dtype]
    cai = getattr(obj, "__cuda_array_interface__", None)
    if cai is None:
        raise TypeError(f"object of type {type(obj)} does not support __cuda_array_interface__")
    return cai["data"][0], int(np.prod(cai["shape"])), _NP_TO_C[np.dtype(cai["typestr"]).name]


class _View:
    """Borrowed cugraph_type_erased_device_array_view_t over a Python device array (utils.pyx:235-246)."""

    def __init__(self, obj):
        self.ptr = None
        if obj is None:
            return
        p, n, t = _describe(obj)
        self._keepalive = obj
        self.ptr = capi.lib().cugraph_type_erased_device_array_view_create(C.c_void_p(p), n, t)

    def free(self):
        if self.ptr:
            capi.lib().cugraph_type_erased_device_array_view_free(self.ptr)
            self.ptr = None


def _sync_torch():
    # inputs may still be in flight on torch's stream; the library runs on its own HIP stream
    torch.cuda.current_stream().synchronize()


def copy_to_torch(handle_ptr, view_ptr):
    """utils.pyx:162-196 copy_to_cupy_array: new tensor <- device view, then free the view."""
    l = capi.lib()
    n = l.cugraph_type_erased_device_array_view_size(view_ptr)
    t = l.cugraph_type_erased_device_array_view_type(view_ptr)
    out = torch.empty(n, dtype=_C_TO_TORCH[t], device="cuda")
    _sync_torch()
    ov = l.cugraph_type_erased_device_array_view_create(C.c_void_p(out.data_ptr()), n, t)
    err = C.c_void_p()
    code = l.cugraph_type_erased_device_array_view_copy(handle_ptr, ov, view_ptr, C.byref(err))
    l.cugraph_type_erased_device_array_view_free(ov)
    l.cugraph_type_erased_device_array_view_free(view_ptr)
    assert_success(code, err, "cugraph_type_erased_device_array_view_copy")
    return out


class Comm:
    """The library's one-node communicator (include/cugraph_amd/extensions.h: cugraph_amd_comm_create; csrc/comm.hpp): one process per
    rank, HIP IPC windows, direct peer writes.  Collective constructor: every rank passes the same session name and size."""

    def __init__(self, session: str, rank: int, size: int):
        if not torch.cuda.is_available():
            raise RuntimeError("cugraph_amd needs a HIP device; there is no CPU fallback")
        torch.cuda.init()
        torch.cuda.current_device()
        ptr, err = C.c_void_p(), C.c_void_p()
        assert_success(capi.lib().cugraph_amd_comm_create(session.encode(), int(rank), int(size), C.byref(ptr), C.byref(err)), err, "cugraph_amd_comm_create")
        self.c_comm_ptr = ptr
        self.rank, self.size = int(rank), int(size)

    def barrier(self):
        """host barrier over all ranks (timing harness; the algorithms synchronise on the device)"""
        err = C.c_void_p()
        assert_success(capi.lib().cugraph_amd_comm_host_barrier(self.c_comm_ptr, C.byref(err)), err, "cugraph_amd_comm_host_barrier")

    def allgather_f64(self, values):
        """every rank's list of doubles (same length everywhere, at most 512) -> list of lists, in rank order"""
        n = len(values)
        mine = (C.c_double * n)(*[float(v) for v in values])
        out = (C.c_double * (n * self.size))()
        err = C.c_void_p()
        assert_success(capi.lib().cugraph_amd_comm_host_allgather(self.c_comm_ptr, mine, 8 * n, out, C.byref(err)), err, "cugraph_amd_comm_host_allgather")
        return [[out[r * n + k] for k in range(n)] for r in range(self.size)]

    def close(self):
        p = getattr(self, "c_comm_ptr", None)
        if p:
            capi.lib().cugraph_amd_comm_free(p)
            self.c_comm_ptr = None

    def __del__(self):
        self.close()


class ResourceHandle:
    """resource_handle.pyx: RAII wrapper of cugraph_resource_handle_t (SG: library-owned context)."""

    def __init__(self, handle=None):
        """handle: None (library-owned single-GPU context) or a Comm (the multi-GPU communicator; the reference passes the address of a
        raft::handle_t that carries NCCL comms here, resource_handle.pyx:47-66) -- an integer is taken as that pointer."""
        l = capi.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("cugraph_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
        torch.cuda.init()
        self.comm = handle if isinstance(handle, Comm) else None
        ptr = None if handle is None else C.c_void_p(handle.c_comm_ptr.value if isinstance(handle, Comm) else int(handle))
        self.c_resource_handle_ptr = l.cugraph_create_resource_handle(ptr)
        if not self.c_resource_handle_ptr:
            raise RuntimeError("cugraph_create_resource_handle failed")

    @property
    def rank(self):
        return capi.lib().cugraph_resource_handle_get_rank(self.c_resource_handle_ptr)

    @property
    def comm_size(self):
        return capi.lib().cugraph_resource_handle_get_comm_size(self.c_resource_handle_ptr)

    def __del__(self):
        p = getattr(self, "c_resource_handle_ptr", None)
        if p and _lib_at_exit() is not None:
            _lib_at_exit().cugraph_free_resource_handle(p)
            self.c_resource_handle_ptr = None

    # --- harness helpers (extensions.h)
    def set_stream(self, hip_stream):
        """Share a caller's HIP stream (an integer handle, e.g. torch.cuda.current_stream().cuda_stream; None = own stream)."""
        err = C.c_void_p()
        assert_success(capi.lib().cugraph_amd_handle_set_stream(self.c_resource_handle_ptr, C.c_void_p(hip_stream) if hip_stream else None, C.byref(err)),
                       err, "cugraph_amd_handle_set_stream")

    def sync(self):
        err = C.c_void_p()
        assert_success(capi.lib().cugraph_amd_handle_sync(self.c_resource_handle_ptr, C.byref(err)), err, "cugraph_amd_handle_sync")

    def kernel_timing(self, on: bool):
        capi.lib().cugraph_amd_kernel_timing_enable(self.c_resource_handle_ptr, 1 if on else 0)

    def kernel_timing_region(self, family: str, begin: bool):
        """one HIP-event pair around a region of work on the handle's stream (read it with kernel_timing_get(family))"""
        l = capi.lib()
        (l.cugraph_amd_kernel_timing_region_begin if begin else l.cugraph_amd_kernel_timing_region_end)(self.c_resource_handle_ptr, family.encode())

    def kernel_timing_reset(self):
        capi.lib().cugraph_amd_kernel_timing_reset(self.c_resource_handle_ptr)

    def kernel_timing_get(self, family: str):
        n, ms, err = C.c_size_t(0), C.c_double(0), C.c_void_p()
        assert_success(capi.lib().cugraph_amd_kernel_timing_get(self.c_resource_handle_ptr, family.encode(), C.byref(n), C.byref(ms),
                                                                C.byref(err)), err, "cugraph_amd_kernel_timing_get")
        return int(n.value), float(ms.value)

    def set_pagerank_hot_tile(self, n_entries: int) -> int:
        return capi.lib().cugraph_amd_set_pagerank_hot_tile(self.c_resource_handle_ptr, int(n_entries))

    def last_traversal_stats(self):
        s = capi.TraversalStats()
        capi.lib().cugraph_amd_last_traversal_stats(self.c_resource_handle_ptr, C.byref(s))
        return {"steps": s.steps, "edges_inspected": s.edges_inspected, "vertices_reached": s.vertices_reached,
                "edges_of_reached": s.edges_of_reached, "probes": s.probes}


class GraphProperties:
    """graph_properties.pyx"""

    def __init__(self, is_symmetric=False, is_multigraph=False):
        self.c_graph_properties = capi.GraphPropertiesStruct(int(bool(is_symmetric)), int(bool(is_multigraph)))

    @property
    def is_symmetric(self):
        return bool(self.c_graph_properties.is_symmetric)

    @property
    def is_multigraph(self):
        return bool(self.c_graph_properties.is_multigraph)


class SGGraph:
    """graphs.pyx:42-337 SGGraph: owns a cugraph_graph_t created by cugraph_graph_create_with_times_sg
    (COO) or cugraph_graph_create_sg_from_csr (CSR)."""

    def __init__(self, resource_handle, graph_properties, src_or_offset_array, dst_or_index_array, weight_array=None,
                 store_transposed=False, renumber=False, do_expensive_check=False, edge_id_array=None, edge_type_array=None,
                 edge_start_time_array=None, edge_end_time_array=None, input_array_format="COO", vertices_array=None,
                 drop_self_loops=False, drop_multi_edges=False, symmetrize=False):
        self.c_graph_ptr = None
        l = capi.lib()
        for name, arr in (("src_or_offset_array", src_or_offset_array), ("dst_or_index_array", dst_or_index_array)):
            _describe(arr)  # TypeError like assert_CAI_type
        self.resource_handle = resource_handle
        views = [_View(a) for a in (vertices_array, src_or_offset_array, dst_or_index_array, weight_array, edge_id_array,
                                    edge_type_array, edge_start_time_array, edge_end_time_array)]
        v_vert, v_src, v_dst, v_w, v_eid, v_et, v_t0, v_t1 = [v.ptr for v in views]
        g, err = C.c_void_p(), C.c_void_p()
        _sync_torch()
        try:
            if input_array_format == "COO":
                code = l.cugraph_graph_create_with_times_sg(
                    resource_handle.c_resource_handle_ptr, C.byref(graph_properties.c_graph_properties), v_vert, v_src, v_dst, v_w,
                    v_eid, v_et, v_t0, v_t1, int(store_transposed), int(renumber), int(drop_self_loops), int(drop_multi_edges),
                    int(symmetrize), int(do_expensive_check), C.byref(g), C.byref(err))
                assert_success(code, err, "cugraph_graph_create_with_times_sg()")
            elif input_array_format == "CSR":
                code = l.cugraph_graph_create_sg_from_csr(
                    resource_handle.c_resource_handle_ptr, C.byref(graph_properties.c_graph_properties), v_src, v_dst, v_w, v_eid,
                    v_et, int(store_transposed), int(renumber), int(symmetrize), int(do_expensive_check), C.byref(g), C.byref(err))
                assert_success(code, err, "cugraph_graph_create_sg_from_csr()")
            else:
                raise ValueError("invalid 'input_array_format'. Only 'COO' and 'CSR' format are supported.")
        finally:
            for v in views:
                v.free()
        self.c_graph_ptr = g

    def __del__(self):
        p = getattr(self, "c_graph_ptr", None)
        if p and _lib_at_exit() is not None:
            _lib_at_exit().cugraph_graph_free(p)
            self.c_graph_ptr = None

    @property
    def num_vertices(self):
        return int(capi.lib().cugraph_amd_graph_num_vertices(self.c_graph_ptr))

    @property
    def num_edges(self):
        return int(capi.lib().cugraph_amd_graph_num_edges(self.c_graph_ptr))

    def compress_hypersparse(self, transposed, first_row=0):
        """extensions.h cugraph_amd_graph_compress_hypersparse: rows >= first_row of the orientation keep an offset only when they have an edge
        (the reference's CSR + DCSR hybrid, structure_utils.cuh:139-195)"""
        err = C.c_void_p()
        _sync_torch()
        assert_success(capi.lib().cugraph_amd_graph_compress_hypersparse(self.resource_handle.c_resource_handle_ptr, self.c_graph_ptr, int(transposed),
                                                                         int(first_row), C.byref(err)), err, "cugraph_amd_graph_compress_hypersparse")

    def hypersparse_view(self, transposed):
        """(is_hypersparse, first_row, nzd_rows, offsets): the row storage of the orientation as it is now (copies of the device arrays)"""
        l = capi.lib()
        h = self.resource_handle.c_resource_handle_ptr
        flag, first, n_nzd = C.c_int(0), C.c_size_t(0), C.c_size_t(0)
        nzd, off, err = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _sync_torch()
        assert_success(l.cugraph_amd_graph_hypersparse_view(h, self.c_graph_ptr, int(transposed), C.byref(flag), C.byref(first), C.byref(n_nzd), C.byref(nzd),
                                                            C.byref(off), C.byref(err)), err, "cugraph_amd_graph_hypersparse_view")
        return bool(flag.value), int(first.value), copy_to_torch(h, nzd), copy_to_torch(h, off)  # (copy_to_torch frees the borrowed views)

    def num_local_edges(self):
        """multi-GPU graph: edges of this rank's PageRank partition (0 before the first PageRank call); otherwise all edges"""
        return int(capi.lib().cugraph_amd_graph_num_local_edges(self.c_graph_ptr))


class MGGraph(SGGraph):
    """graphs.pyx:357-700 MGGraph: this rank's slice of the edge list, as one array per column or as `num_arrays` lists of arrays
    (cugraph_graph_create_with_times_mg concatenates them).  On a ResourceHandle created on a Comm the call is COLLECTIVE: the slices of
    all ranks become one partitioned graph (csrc/mg_graph.hip) that pagerank / personalized_pagerank / bfs / sssp / louvain / degrees /
    has_vertex / bfs_extract_paths / decompress_to_edgelist accept, each rank getting its share of the vertices (or of the edge list) back; vertex
    ids may be int32 or int64 (the ranks agree on one sorted id list), edge ids / edge type ids stay with the rank's slice.  On a plain handle (one rank, no communicator) the graph is built as SGGraph builds it, always renumbered
    (graph_mg.cpp:214)."""

    def __init__(self, resource_handle, graph_properties, src_array, dst_array, weight_array=None, store_transposed=False,
                 do_expensive_check=False, edge_id_array=None, edge_type_array=None, edge_start_time_array=None, edge_end_time_array=None,
                 vertices_array=None, num_arrays=1, drop_self_loops=False, drop_multi_edges=False, symmetrize=False):
        self.c_graph_ptr = None
        l = capi.lib()
        self.resource_handle = resource_handle

        def as_list(a, name):
            if a is None:
                return None
            seq = list(a) if isinstance(a, (list, tuple)) else [a]
            if len(seq) != num_arrays:
                raise ValueError(f"{name}: {len(seq)} arrays given, num_arrays = {num_arrays}")
            for x in seq:
                _describe(x)  # TypeError like assert_CAI_type
            return seq

        cols = [as_list(a, n) for a, n in ((vertices_array, "vertices_array"), (src_array, "src_array"), (dst_array, "dst_array"),
                                           (weight_array, "weight_array"), (edge_id_array, "edge_id_array"), (edge_type_array, "edge_type_array"),
                                           (edge_start_time_array, "edge_start_time_array"), (edge_end_time_array, "edge_end_time_array"))]
        if cols[1] is None or cols[2] is None:
            raise TypeError("src_array and dst_array are required")
        views, ptr_arrays = [], []
        for seq in cols:
            if seq is None:
                ptr_arrays.append(None)
                continue
            vs = [_View(x) for x in seq]
            views += vs
            ptr_arrays.append((C.c_void_p * num_arrays)(*[v.ptr for v in vs]))
        g, err = C.c_void_p(), C.c_void_p()
        _sync_torch()
        try:
            code = l.cugraph_graph_create_with_times_mg(
                resource_handle.c_resource_handle_ptr, C.byref(graph_properties.c_graph_properties), *ptr_arrays, int(store_transposed),
                int(num_arrays), int(drop_self_loops), int(drop_multi_edges), int(symmetrize), int(do_expensive_check), C.byref(g), C.byref(err))
            assert_success(code, err, "cugraph_graph_create_with_times_mg()")
        finally:
            for v in views:
                v.free()
        self.c_graph_ptr = g


def has_vertex(resource_handle, graph, vertices, do_expensive_check=False):
    """has_vertex.pyx:43"""
    l = capi.lib()
    v = _View(vertices)
    res, err = C.c_void_p(), C.c_void_p()
    _sync_torch()
    code = l.cugraph_has_vertex(resource_handle.c_resource_handle_ptr, graph.c_graph_ptr, v.ptr, int(do_expensive_check),
                                C.byref(res), C.byref(err))
    v.free()
    assert_success(code, err, "cugraph_has_vertex")
    view = l.cugraph_type_erased_device_array_view(res)
    out = copy_to_torch(resource_handle.c_resource_handle_ptr, view)
    l.cugraph_type_erased_device_array_free(res)
    return out


def _centrality_result(resource_handle, result_ptr, fail_on_nonconvergence):
    l = capi.lib()
    converged = bool(l.cugraph_centrality_result_converged(result_ptr))
    verts = vals = None
    if (fail_on_nonconvergence is False) or converged:
        verts = copy_to_torch(resource_handle.c_resource_handle_ptr, l.cugraph_centrality_result_get_vertices(result_ptr))
        vals = copy_to_torch(resource_handle.c_resource_handle_ptr, l.cugraph_centrality_result_get_values(result_ptr))
    l.cugraph_centrality_result_free(result_ptr)
    if fail_on_nonconvergence is False:
        return (verts, vals, converged)
    if converged:
        return (verts, vals)
    raise FailedToConvergeError


def pagerank(resource_handle, graph, precomputed_vertex_out_weight_vertices, precomputed_vertex_out_weight_sums,
             initial_guess_vertices, initial_guess_values, alpha, epsilon, max_iterations, do_expensive_check,
             fail_on_nonconvergence=True):
    """pagerank.pyx:49-237 (always calls cugraph_pagerank_allow_nonconvergence, :196)."""
    l = capi.lib()
    views = [_View(a) for a in (precomputed_vertex_out_weight_vertices, precomputed_vertex_out_weight_sums,
                                initial_guess_vertices, initial_guess_values)]
    res, err = C.c_void_p(), C.c_void_p()
    _sync_torch()
    code = l.cugraph_pagerank_allow_nonconvergence(resource_handle.c_resource_handle_ptr, graph.c_graph_ptr, views[0].ptr, views[1].ptr,
                                                   views[2].ptr, views[3].ptr, float(alpha), float(epsilon), int(max_iterations),
                                                   int(do_expensive_check), C.byref(res), C.byref(err))
    for v in views:
        v.free()
    assert_success(code, err, "cugraph_pagerank_allow_nonconvergence")
    return _centrality_result(resource_handle, res, fail_on_nonconvergence)


def personalized_pagerank(resource_handle, graph, precomputed_vertex_out_weight_vertices, precomputed_vertex_out_weight_sums,
                          initial_guess_vertices, initial_guess_values, personalization_vertices, personalization_values, alpha,
                          epsilon, max_iterations, do_expensive_check, fail_on_nonconvergence=True):
    """personalized_pagerank.pyx:49-264"""
    l = capi.lib()
    views = [_View(a) for a in (precomputed_vertex_out_weight_vertices, precomputed_vertex_out_weight_sums, initial_guess_vertices,
                                initial_guess_values, personalization_vertices, personalization_values)]
    res, err = C.c_void_p(), C.c_void_p()
    _sync_torch()
    code = l.cugraph_personalized_pagerank_allow_nonconvergence(
        resource_handle.c_resource_handle_ptr, graph.c_graph_ptr, *[v.ptr for v in views], float(alpha), float(epsilon),
        int(max_iterations), int(do_expensive_check), C.byref(res), C.byref(err))
    for v in views:
        v.free()
    assert_success(code, err, "cugraph_personalized_pagerank_allow_nonconvergence")
    return _centrality_result(resource_handle, res, fail_on_nonconvergence)


# wall time of the latest blocking C call per entry point (seconds): what the C-ABI boundary itself took, without this mirror's pre-checks and result
# copies (bench_traversal.py reports both)
last_c_call_s = {}


def bfs(handle, graph, sources, direction_optimizing, depth_limit, compute_predecessors, do_expensive_check):
    """bfs.pyx:50-196.  Returns (distances, predecessors, vertices)."""
    l = capi.lib()
    _describe(sources)
    if not bool(torch.all(has_vertex(handle, graph, sources, do_expensive_check))):  # bfs.pyx:140-143
        raise ValueError("one or more vertices are invalid. Call the method 'has_vertex' ", "to identify the invalid vertices")
    if depth_limit <= 0:
        depth_limit = INT_MAX - 1  # bfs.pyx:144-145
    v = _View(sources)
    res, err = C.c_void_p(), C.c_void_p()
    _sync_torch()
    t0 = time.perf_counter()
    code = l.cugraph_bfs(handle.c_resource_handle_ptr, graph.c_graph_ptr, v.ptr, int(direction_optimizing), int(depth_limit),
                         int(compute_predecessors), int(do_expensive_check), C.byref(res), C.byref(err))
    last_c_call_s["cugraph_bfs"] = time.perf_counter() - t0  # (the call returns when its results are complete)
    v.free()
    assert_success(code, err, "cugraph_bfs")
    h = handle.c_resource_handle_ptr
    distances = copy_to_torch(h, l.cugraph_paths_result_get_distances(res))
    predecessors = copy_to_torch(h, l.cugraph_paths_result_get_predecessors(res))
    vertices = copy_to_torch(h, l.cugraph_paths_result_get_vertices(res))
    l.cugraph_paths_result_free(res)
    return (distances, predecessors, vertices)


def bfs_extract_paths(handle, graph, sources, destinations, direction_optimizing=False, depth_limit=0):
    """cugraph_bfs with predecessors followed by cugraph_extract_paths (traversal_algorithms.h:167-201) on the same result,
    the way cpp/tests/c_api/extract_paths_test.c drives it.  Returns (distances, predecessors, vertices, paths) with paths a
    (len(destinations), max_path_length) int32 matrix of external ids padded with -1."""
    l = capi.lib()
    _describe(sources)
    _describe(destinations)
    if depth_limit <= 0:
        depth_limit = INT_MAX - 1
    v, dv = _View(sources), _View(destinations)
    res, ext, err = C.c_void_p(), C.c_void_p(), C.c_void_p()
    _sync_torch()
    h = handle.c_resource_handle_ptr
    try:
        code = l.cugraph_bfs(h, graph.c_graph_ptr, v.ptr, int(direction_optimizing), int(depth_limit), 1, 0, C.byref(res), C.byref(err))
        assert_success(code, err, "cugraph_bfs")
        code = l.cugraph_extract_paths(h, graph.c_graph_ptr, v.ptr, res, dv.ptr, C.byref(ext), C.byref(err))
        assert_success(code, err, "cugraph_extract_paths")
        distances = copy_to_torch(h, l.cugraph_paths_result_get_distances(res))
        predecessors = copy_to_torch(h, l.cugraph_paths_result_get_predecessors(res))
        vertices = copy_to_torch(h, l.cugraph_paths_result_get_vertices(res))
        length = int(l.cugraph_extract_paths_result_get_max_path_length(ext))
        paths = copy_to_torch(h, l.cugraph_extract_paths_result_get_paths(ext)).reshape(-1, length)
    finally:
        v.free()
        dv.free()
        if ext:
            l.cugraph_extract_paths_result_free(ext)
        if res:
            l.cugraph_paths_result_free(res)
    return (distances, predecessors, vertices, paths)


def decompress_to_edgelist(resource_handle, graph, do_expensive_check=False):
    """decompress_to_edgelist.pyx: the graph back as (sources, destinations, weights, edge_ids, edge_type_ids) with the caller's vertex ids (missing
    columns None).  On a multi-GPU graph every rank gets ITS part of the edge list (not collective)."""
    l = capi.lib()
    h = resource_handle.c_resource_handle_ptr
    el, err = C.c_void_p(), C.c_void_p()
    _sync_torch()
    assert_success(l.cugraph_decompress_to_edgelist(h, graph.c_graph_ptr, int(do_expensive_check), C.byref(el), C.byref(err)), err, "cugraph_decompress_to_edgelist")
    try:
        cols = []
        for get in (l.cugraph_edgelist_get_sources, l.cugraph_edgelist_get_destinations, l.cugraph_edgelist_get_edge_weights, l.cugraph_edgelist_get_edge_ids,
                    l.cugraph_edgelist_get_edge_type_ids):
            view = get(el)
            cols.append(copy_to_torch(h, view) if view else None)
    finally:
        l.cugraph_edgelist_free(el)
    return tuple(cols)


def louvain(resource_handle, graph, max_level, threshold, resolution, do_expensive_check):
    """louvain.pyx: returns (vertices, clusters, modularity)."""
    l = capi.lib()
    res, err = C.c_void_p(), C.c_void_p()
    _sync_torch()
    code = l.cugraph_louvain(resource_handle.c_resource_handle_ptr, graph.c_graph_ptr, int(max_level), float(threshold), float(resolution),
                             int(do_expensive_check), C.byref(res), C.byref(err))
    assert_success(code, err, "cugraph_louvain")
    h = resource_handle.c_resource_handle_ptr
    vertices = copy_to_torch(h, l.cugraph_hierarchical_clustering_result_get_vertices(res))
    clusters = copy_to_torch(h, l.cugraph_hierarchical_clustering_result_get_clusters(res))
    modularity = float(l.cugraph_hierarchical_clustering_result_get_modularity(res))
    l.cugraph_hierarchical_clustering_result_free(res)
    return (vertices, clusters, modularity)


def _degrees(name, resource_handle, graph, source_vertices, do_expensive_check, want_in, want_out):
    l = capi.lib()
    sv = None
    if source_vertices is not None:
        _describe(source_vertices)
        sv = _View(source_vertices)
    res, err = C.c_void_p(), C.c_void_p()
    _sync_torch()
    code = getattr(l, name)(resource_handle.c_resource_handle_ptr, graph.c_graph_ptr, (sv.ptr if sv else None), int(do_expensive_check),
                            C.byref(res), C.byref(err))
    if sv:
        sv.free()
    assert_success(code, err, name)
    h = resource_handle.c_resource_handle_ptr
    out = [copy_to_torch(h, l.cugraph_degrees_result_get_vertices(res))]
    if want_in:
        out.append(copy_to_torch(h, l.cugraph_degrees_result_get_in_degrees(res)))
    if want_out:
        out.append(copy_to_torch(h, l.cugraph_degrees_result_get_out_degrees(res)))
    l.cugraph_degrees_result_free(res)
    return tuple(out)


def in_degrees(resource_handle, graph, source_vertices=None, do_expensive_check=False):
    """degrees.pyx in_degrees: (vertices, in_degrees)."""
    return _degrees("cugraph_in_degrees", resource_handle, graph, source_vertices, do_expensive_check, True, False)


def out_degrees(resource_handle, graph, source_vertices=None, do_expensive_check=False):
    """degrees.pyx out_degrees: (vertices, out_degrees)."""
    return _degrees("cugraph_out_degrees", resource_handle, graph, source_vertices, do_expensive_check, False, True)


def degrees(resource_handle, graph, source_vertices=None, do_expensive_check=False):
    """degrees.pyx degrees: (vertices, in_degrees, out_degrees)."""
    return _degrees("cugraph_degrees", resource_handle, graph, source_vertices, do_expensive_check, True, True)


def sssp(resource_handle, graph, source, cutoff, compute_predecessors, do_expensive_check):
    """sssp.pyx:48-168.  Returns (vertices, distances, predecessors)."""
    l = capi.lib()
    res, err = C.c_void_p(), C.c_void_p()
    _sync_torch()
    t0 = time.perf_counter()
    code = l.cugraph_sssp(resource_handle.c_resource_handle_ptr, graph.c_graph_ptr, int(source), float(cutoff), int(compute_predecessors),
                          int(do_expensive_check), C.byref(res), C.byref(err))
    last_c_call_s["cugraph_sssp"] = time.perf_counter() - t0
    assert_success(code, err, "cugraph_sssp")
    h = resource_handle.c_resource_handle_ptr
    vertices = copy_to_torch(h, l.cugraph_paths_result_get_vertices(res))
    distances = copy_to_torch(h, l.cugraph_paths_result_get_distances(res))
    predecessors = copy_to_torch(h, l.cugraph_paths_result_get_predecessors(res))
    l.cugraph_paths_result_free(res)
    return (vertices, distances, predecessors)


# ------------------------------------------------------------------------------- harness extensions
def rmat_edgelist(resource_handle, seed, scale, num_edges, a=0.57, b=0.19, c=0.19, clip_and_flip=False, scramble_vertex_ids=False,
                  include_edge_weights=False, minimum_weight=0.0, maximum_weight=1.0, dtype=None):
    """The reference's generator API (pylibcugraph generate_rmat_edgelist.pyx: cugraph_rng_state_create,
    cugraph_generate_rmat_edgelist, cugraph_generate_edge_weights, cugraph_coo_*).  Returns (src, dst, weights or None)."""
    l = capi.lib()
    h = resource_handle.c_resource_handle_ptr
    rng, coo, err = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert_success(l.cugraph_rng_state_create(h, int(seed), C.byref(rng), C.byref(err)), err, "cugraph_rng_state_create")
    try:
        code = l.cugraph_generate_rmat_edgelist(h, rng, int(scale), int(num_edges), float(a), float(b), float(c), int(clip_and_flip),
                                                int(scramble_vertex_ids), C.byref(coo), C.byref(err))
        assert_success(code, err, "cugraph_generate_rmat_edgelist")
        w = None
        if include_edge_weights:
            t = capi.FLOAT64 if dtype in (torch.float64, "float64") else capi.FLOAT32
            code = l.cugraph_generate_edge_weights(h, rng, coo, t, float(minimum_weight), float(maximum_weight), C.byref(err))
            assert_success(code, err, "cugraph_generate_edge_weights")
            w = copy_to_torch(h, l.cugraph_coo_get_edge_weights(coo))
        src = copy_to_torch(h, l.cugraph_coo_get_sources(coo))
        dst = copy_to_torch(h, l.cugraph_coo_get_destinations(coo))
        assert not l.cugraph_coo_get_edge_id(coo) and not l.cugraph_coo_get_edge_type(coo)
    finally:
        if coo:
            l.cugraph_coo_free(coo)
        l.cugraph_rng_state_free(rng)
    return src, dst, w


def read_matrix_market(resource_handle, path):
    """MatrixMarket coordinate file -> (src, dst, weights, num_vertices, is_symmetric, has_weights) on the device
    (cugraph_amd_read_matrix_market; conventions of the reference's test reader, matrix_market_file_utilities.cu)."""
    l = capi.lib()
    h = resource_handle.c_resource_handle_ptr
    coo, err = C.c_void_p(), C.c_void_p()
    nv, sym, hw = C.c_size_t(0), C.c_int(0), C.c_int(0)
    code = l.cugraph_amd_read_matrix_market(h, str(path).encode(), C.byref(coo), C.byref(nv), C.byref(sym), C.byref(hw), C.byref(err))
    assert_success(code, err, "cugraph_amd_read_matrix_market")
    try:
        src = copy_to_torch(h, l.cugraph_coo_get_sources(coo))
        dst = copy_to_torch(h, l.cugraph_coo_get_destinations(coo))
        w = copy_to_torch(h, l.cugraph_coo_get_edge_weights(coo))
    finally:
        l.cugraph_coo_free(coo)
    return src, dst, w, int(nv.value), bool(sym.value), bool(hw.value)


def generate_rmat_edgelist(resource_handle, scale, num_edges, a=0.57, b=0.19, c=0.19, seed=0, first_edge=0):
    """On-device RMAT slice [first_edge, first_edge + num_edges) -> (src, dst) int32 tensors."""
    l = capi.lib()
    src = torch.empty(num_edges, dtype=torch.int32, device="cuda")
    dst = torch.empty(num_edges, dtype=torch.int32, device="cuda")
    vs, vd = _View(src), _View(dst)
    err = C.c_void_p()
    _sync_torch()
    code = l.cugraph_amd_generate_rmat_edgelist(resource_handle.c_resource_handle_ptr, int(scale), int(first_edge), int(num_edges),
                                                float(a), float(b), float(c), int(seed), vs.ptr, vd.ptr, C.byref(err))
    vs.free()
    vd.free()
    assert_success(code, err, "cugraph_amd_generate_rmat_edgelist")
    return src, dst


class PageRankPlan:
    """Explicit create / step / result form of cugraph_pagerank (extensions.h) so a harness can time
    exactly K power iterations."""

    def __init__(self, resource_handle, graph, alpha=0.85, initial_guess=None, personalization=None, precomputed_out_weights=None):
        l = capi.lib()
        self.handle, self.graph = resource_handle, graph
        pairs = []
        for p in (precomputed_out_weights, initial_guess, personalization):
            pairs += [_View(p[0]), _View(p[1])] if p is not None else [_View(None), _View(None)]
        plan, err = C.c_void_p(), C.c_void_p()
        _sync_torch()
        code = l.cugraph_amd_pagerank_plan_create(resource_handle.c_resource_handle_ptr, graph.c_graph_ptr, *[v.ptr for v in pairs],
                                                  float(alpha), C.byref(plan), C.byref(err))
        for v in pairs:
            v.free()
        assert_success(code, err, "cugraph_amd_pagerank_plan_create")
        self.ptr = plan
        self.iterations = 0

    def tune(self, placements=8):
        """extensions.h cugraph_amd_pagerank_plan_tune: before the first step, time the plan on `placements` placements of its streamed arrays and keep
        the fastest; returns the kept placement's milliseconds per iteration (0.0: nothing to tune)"""
        ms, err = C.c_double(0.0), C.c_void_p()
        assert_success(capi.lib().cugraph_amd_pagerank_plan_tune(self.ptr, int(placements), C.byref(ms), C.byref(err)), err, "cugraph_amd_pagerank_plan_tune")
        return float(ms.value)

    def step(self, n_iterations, epsilon=0.0):
        done, conv, err = C.c_size_t(0), C.c_int(0), C.c_void_p()
        code = capi.lib().cugraph_amd_pagerank_plan_step(self.ptr, float(epsilon), int(n_iterations), C.byref(done), C.byref(conv), C.byref(err))
        assert_success(code, err, "cugraph_amd_pagerank_plan_step")
        self.iterations += int(done.value)
        return int(done.value), bool(conv.value)

    def result(self, converged=False):
        l = capi.lib()
        res, err = C.c_void_p(), C.c_void_p()
        code = l.cugraph_amd_pagerank_plan_result(self.ptr, self.iterations, int(converged), C.byref(res), C.byref(err))
        assert_success(code, err, "cugraph_amd_pagerank_plan_result")
        return _centrality_result(self.handle, res, False)

    def __del__(self):
        p = getattr(self, "ptr", None)
        if p and _lib_at_exit() is not None:
            _lib_at_exit().cugraph_amd_pagerank_plan_free(p)
            self.ptr = None
