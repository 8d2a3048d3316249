"""ctypes binding of the C-ABI library (cugraph_amd/lib/libcugraph_c.so, headers in include/).

This is the same boundary pylibcugraph's Cython modules bind (python/pylibcugraph/pylibcugraph/_cugraph_c/*.pxd);
see INTEGRATION.md.  There is NO CPU fallback: if the library is missing or cannot be loaded the import
of anything that computes fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_DIR = _PKG / "lib"
LIB_PATH = LIB_DIR / "libcugraph_c.so"
CSRC = _PKG / "csrc"

# cugraph_error_code_t (include/cugraph_c/error.h)
CUGRAPH_SUCCESS, CUGRAPH_UNKNOWN_ERROR, CUGRAPH_INVALID_HANDLE, CUGRAPH_ALLOC_ERROR, CUGRAPH_INVALID_INPUT, \
    CUGRAPH_NOT_IMPLEMENTED, CUGRAPH_UNSUPPORTED_TYPE_COMBINATION = range(7)
ERROR_NAMES = ["CUGRAPH_SUCCESS", "CUGRAPH_UNKNOWN_ERROR", "CUGRAPH_INVALID_HANDLE", "CUGRAPH_ALLOC_ERROR",
               "CUGRAPH_INVALID_INPUT", "CUGRAPH_NOT_IMPLEMENTED", "CUGRAPH_UNSUPPORTED_TYPE_COMBINATION"]
# cugraph_data_type_id_t (include/cugraph_c/types.h)
INT8, INT16, INT32, INT64, UINT8, UINT16, UINT32, UINT64, FLOAT32, FLOAT64, SIZE_T, BOOL, NTYPES = range(13)


class GraphPropertiesStruct(C.Structure):
    _fields_ = [("is_symmetric", C.c_int), ("is_multigraph", C.c_int)]


class TraversalStats(C.Structure):
    _fields_ = [("steps", C.c_uint64), ("edges_inspected", C.c_uint64), ("vertices_reached", C.c_uint64),
                ("edges_of_reached", C.c_uint64), ("probes", C.c_uint64)]


def build(verbose: bool = False) -> Path:
    """Compile every HIP source for gfx950 into cugraph_amd/lib/ (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", str(CSRC), "-j", str(max(1, min(8, os.cpu_count() or 1)))]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libcugraph_c.so failed (see output above)")
    return LIB_PATH


_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); every symbol declared in include/cugraph_c/*.h and include/cugraph_amd/extensions.h
PROTOTYPES = {
    # error.h
    "cugraph_error_message": (C.c_char_p, [_P]),
    "cugraph_error_free": (None, [_P]),
    # resource_handle.h
    "cugraph_create_resource_handle": (_P, [_P]),
    "cugraph_resource_handle_get_comm_size": (C.c_int, [_P]),
    "cugraph_resource_handle_get_rank": (C.c_int, [_P]),
    "cugraph_free_resource_handle": (None, [_P]),
    # array.h
    "cugraph_type_erased_device_array_create": (C.c_int, [_P, C.c_size_t, C.c_int, _PP, _PP]),
    "cugraph_type_erased_device_array_create_from_view": (C.c_int, [_P, _P, _PP, _PP]),
    "cugraph_type_erased_device_array_free": (None, [_P]),
    "cugraph_type_erased_device_array_release": (_P, [_P]),
    "cugraph_type_erased_device_array_view": (_P, [_P]),
    "cugraph_type_erased_device_array_view_as_type": (C.c_int, [_P, C.c_int, _PP, _PP]),
    "cugraph_type_erased_device_array_view_create": (_P, [_P, C.c_size_t, C.c_int]),
    "cugraph_type_erased_device_array_view_free": (None, [_P]),
    "cugraph_type_erased_device_array_view_size": (C.c_size_t, [_P]),
    "cugraph_type_erased_device_array_view_type": (C.c_int, [_P]),
    "cugraph_type_erased_device_array_view_pointer": (_P, [_P]),
    "cugraph_type_erased_host_array_create": (C.c_int, [_P, C.c_size_t, C.c_int, _PP, _PP]),
    "cugraph_type_erased_host_array_free": (None, [_P]),
    "cugraph_type_erased_host_array_release": (_P, [_P]),
    "cugraph_type_erased_host_array_view": (_P, [_P]),
    "cugraph_type_erased_host_array_view_create": (_P, [_P, C.c_size_t, C.c_int]),
    "cugraph_type_erased_host_array_view_free": (None, [_P]),
    "cugraph_type_erased_host_array_size": (C.c_size_t, [_P]),
    "cugraph_type_erased_host_array_type": (C.c_int, [_P]),
    "cugraph_type_erased_host_array_pointer": (_P, [_P]),
    "cugraph_type_erased_host_array_view_copy": (C.c_int, [_P, _P, _P, _PP]),
    "cugraph_type_erased_device_array_view_copy_from_host": (C.c_int, [_P, _P, _P, _PP]),
    "cugraph_type_erased_device_array_view_copy_to_host": (C.c_int, [_P, _P, _P, _PP]),
    "cugraph_type_erased_device_array_view_copy": (C.c_int, [_P, _P, _P, _PP]),
    # graph.h
    "cugraph_graph_create_sg": (C.c_int, [_P, C.POINTER(GraphPropertiesStruct), _P, _P, _P, _P, _P, _P] + [C.c_int] * 6 + [_PP, _PP]),
    "cugraph_graph_create_with_times_sg": (C.c_int, [_P, C.POINTER(GraphPropertiesStruct), _P, _P, _P, _P, _P, _P, _P, _P] + [C.c_int] * 6 + [_PP, _PP]),
    "cugraph_graph_create_sg_from_csr": (C.c_int, [_P, C.POINTER(GraphPropertiesStruct), _P, _P, _P, _P, _P] + [C.c_int] * 4 + [_PP, _PP]),
    "cugraph_graph_create_mg": (C.c_int, [_P, C.POINTER(GraphPropertiesStruct), _PP, _PP, _PP, _PP, _PP, _PP, C.c_int, C.c_size_t] + [C.c_int] * 4 + [_PP, _PP]),
    "cugraph_graph_create_with_times_mg": (C.c_int, [_P, C.POINTER(GraphPropertiesStruct), _PP, _PP, _PP, _PP, _PP, _PP, _PP, _PP, C.c_int, C.c_size_t] + [C.c_int] * 4 + [_PP, _PP]),
    "cugraph_graph_free": (None, [_P]),
    # graph_functions.h
    "cugraph_has_vertex": (C.c_int, [_P, _P, _P, C.c_int, _PP, _PP]),
    # centrality_algorithms.h
    "cugraph_centrality_result_get_vertices": (_P, [_P]),
    "cugraph_centrality_result_get_values": (_P, [_P]),
    "cugraph_centrality_result_get_num_iterations": (C.c_size_t, [_P]),
    "cugraph_centrality_result_converged": (C.c_int, [_P]),
    "cugraph_centrality_result_free": (None, [_P]),
    "cugraph_pagerank": (C.c_int, [_P] * 6 + [C.c_double, C.c_double, C.c_size_t, C.c_int, _PP, _PP]),
    "cugraph_pagerank_allow_nonconvergence": (C.c_int, [_P] * 6 + [C.c_double, C.c_double, C.c_size_t, C.c_int, _PP, _PP]),
    "cugraph_personalized_pagerank": (C.c_int, [_P] * 8 + [C.c_double, C.c_double, C.c_size_t, C.c_int, _PP, _PP]),
    "cugraph_personalized_pagerank_allow_nonconvergence": (C.c_int, [_P] * 8 + [C.c_double, C.c_double, C.c_size_t, C.c_int, _PP, _PP]),
    # traversal_algorithms.h
    "cugraph_rng_state_create": (C.c_int, [_P, C.c_uint64, _PP, _PP]),
    "cugraph_rng_state_free": (None, [_P]),
    "cugraph_generate_rmat_edgelist": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, _PP, _PP]),
    "cugraph_generate_edge_weights": (C.c_int, [_P, _P, _P, C.c_int, C.c_double, C.c_double, _PP]),
    "cugraph_coo_get_sources": (_P, [_P]),
    "cugraph_coo_get_destinations": (_P, [_P]),
    "cugraph_coo_get_edge_weights": (_P, [_P]),
    "cugraph_coo_get_edge_id": (_P, [_P]),
    "cugraph_coo_get_edge_type": (_P, [_P]),
    "cugraph_coo_free": (None, [_P]),
    "cugraph_coo_list_size": (C.c_size_t, [_P]),
    "cugraph_coo_list_element": (_P, [_P, C.c_size_t]),
    "cugraph_coo_list_free": (None, [_P]),
    "cugraph_generate_rmat_edgelists": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, _PP, _PP]),
    "cugraph_generate_edge_ids": (C.c_int, [_P, _P, C.c_int, _PP]),
    "cugraph_generate_edge_types": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _PP]),
    "cugraph_data_type_id_from_dlpack": (C.c_int, [_P, C.POINTER(C.c_int), _PP]),
    "cugraph_amd_memory_pool_trim": (C.c_size_t, []),
    "cugraph_amd_memory_pool_trim_large": (C.c_size_t, [C.c_size_t]),
    "cugraph_amd_memory_pool_cached_bytes": (C.c_size_t, []),
    "cugraph_louvain": (C.c_int, [_P, _P, C.c_size_t, C.c_double, C.c_double, C.c_int, _PP, _PP]),
    "cugraph_hierarchical_clustering_result_get_vertices": (_P, [_P]),
    "cugraph_hierarchical_clustering_result_get_clusters": (_P, [_P]),
    "cugraph_hierarchical_clustering_result_get_modularity": (C.c_double, [_P]),
    "cugraph_hierarchical_clustering_result_free": (None, [_P]),
    "cugraph_in_degrees": (C.c_int, [_P, _P, _P, C.c_int, _PP, _PP]),
    "cugraph_out_degrees": (C.c_int, [_P, _P, _P, C.c_int, _PP, _PP]),
    "cugraph_degrees": (C.c_int, [_P, _P, _P, C.c_int, _PP, _PP]),
    "cugraph_decompress_to_edgelist": (C.c_int, [_P, _P, C.c_int, _PP, _PP]),
    "cugraph_edgelist_get_sources": (_P, [_P]),
    "cugraph_edgelist_get_destinations": (_P, [_P]),
    "cugraph_edgelist_get_edge_weights": (_P, [_P]),
    "cugraph_edgelist_get_edge_ids": (_P, [_P]),
    "cugraph_edgelist_get_edge_type_ids": (_P, [_P]),
    "cugraph_edgelist_get_edge_offsets": (_P, [_P]),
    "cugraph_edgelist_free": (None, [_P]),
    "cugraph_degrees_result_get_vertices": (_P, [_P]),
    "cugraph_degrees_result_get_in_degrees": (_P, [_P]),
    "cugraph_degrees_result_get_out_degrees": (_P, [_P]),
    "cugraph_degrees_result_free": (None, [_P]),
    "cugraph_extract_paths": (C.c_int, [_P, _P, _P, _P, _P, _PP, _PP]),
    "cugraph_extract_paths_result_get_max_path_length": (C.c_size_t, [_P]),
    "cugraph_extract_paths_result_get_paths": (_P, [_P]),
    "cugraph_extract_paths_result_free": (None, [_P]),
    "cugraph_paths_result_get_vertices": (_P, [_P]),
    "cugraph_paths_result_get_distances": (_P, [_P]),
    "cugraph_paths_result_get_predecessors": (_P, [_P]),
    "cugraph_paths_result_free": (None, [_P]),
    "cugraph_bfs": (C.c_int, [_P, _P, _P, C.c_int, C.c_size_t, C.c_int, C.c_int, _PP, _PP]),
    "cugraph_sssp": (C.c_int, [_P, _P, C.c_size_t, C.c_double, C.c_int, C.c_int, _PP, _PP]),
    # cugraph_amd/extensions.h
    "cugraph_amd_version": (C.c_char_p, []),
    "cugraph_amd_generate_rmat_edgelist": (C.c_int, [_P, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_uint64, _P, _P, _PP]),
    "cugraph_amd_pagerank_plan_create": (C.c_int, [_P] * 8 + [C.c_double, _PP, _PP]),
    "cugraph_amd_pagerank_plan_step": (C.c_int, [_P, C.c_double, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int), _PP]),
    "cugraph_amd_pagerank_plan_tune": (C.c_int, [_P, C.c_size_t, C.POINTER(C.c_double), _PP]),
    "cugraph_amd_pagerank_plan_result": (C.c_int, [_P, C.c_size_t, C.c_int, _PP, _PP]),
    "cugraph_amd_pagerank_plan_free": (None, [_P]),
    "cugraph_amd_pagerank_mg_plan_create": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t, C.c_int, C.c_int, _P, _P, _P, C.POINTER(C.c_size_t),
                                                    C.POINTER(C.c_size_t), _P, _P, _P, C.c_double, _PP, _PP]),
    "cugraph_amd_pagerank_mg_plan_start": (C.c_int, [_P, _PP]),
    "cugraph_amd_pagerank_mg_plan_reduce_scalars": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), _PP]),
    "cugraph_amd_pagerank_mg_plan_local_step": (C.c_int, [_P, _PP]),
    "cugraph_amd_pagerank_mg_plan_values": (C.c_int, [_P, _P, _PP]),
    "cugraph_amd_pagerank_mg_plan_free": (None, [_P]),
    "cugraph_amd_sort_pairs_u64_u32": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_int, C.c_int, _PP]),
    "cugraph_amd_exclusive_scan_u32": (C.c_int, [_P, _P, _P, C.c_size_t, _PP]),
    "cugraph_amd_pagerank_mg2d_plan_create": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _P, _P, _P, _P, _P, _P, _P, C.c_double, _PP, _PP]),
    "cugraph_amd_pagerank_mg2d_plan_start": (C.c_int, [_P, _PP]),
    "cugraph_amd_pagerank_mg2d_plan_set_scalars": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), _PP]),
    "cugraph_amd_pagerank_mg2d_plan_spmv": (C.c_int, [_P, _PP]),
    "cugraph_amd_pagerank_mg2d_plan_epilogue": (C.c_int, [_P, _PP]),
    "cugraph_amd_pagerank_mg2d_plan_values": (C.c_int, [_P, _P, _PP]),
    "cugraph_amd_pagerank_mg2d_plan_free": (None, [_P]),
    "cugraph_amd_traversal_mg_plan_create": (C.c_int, [_P, _P, _P, _P, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, _P, C.c_int, _P,
                                                     C.c_size_t, _PP, _PP]),
    "cugraph_amd_traversal_mg_plan_reset": (C.c_int, [_P, _P, C.c_size_t, C.c_double, C.c_int, _PP]),
    "cugraph_amd_traversal_mg_plan_expand": (C.c_int, [_P, C.POINTER(C.c_size_t), _PP]),
    "cugraph_amd_traversal_mg_plan_apply": (C.c_int, [_P, _P, C.c_size_t, C.c_uint32, C.POINTER(C.c_size_t), _PP]),
    "cugraph_amd_traversal_mg_plan_frontier_bits": (C.c_int, [_P, _PP, _PP]),
    "cugraph_amd_traversal_mg_plan_merge_visited": (C.c_int, [_P, _P, _PP]),
    "cugraph_amd_traversal_mg_plan_sssp_set_window": (C.c_int, [_P, C.c_double, _PP]),
    "cugraph_amd_traversal_mg_plan_sssp_far_stats": (C.c_int, [_P, C.POINTER(C.c_size_t), C.POINTER(C.c_double), _PP]),
    "cugraph_amd_traversal_mg_plan_sssp_advance": (C.c_int, [_P, C.c_double, C.POINTER(C.c_size_t), _PP]),
    "cugraph_amd_traversal_mg_plan_set_bottom_up": (C.c_int, [_P, _P, _P, _P, _PP]),
    "cugraph_amd_traversal_mg_plan_bottom_up": (C.c_int, [_P, _P, C.c_uint32, C.POINTER(C.c_size_t), _PP]),
    "cugraph_amd_traversal_mg_plan_last_degree_sums": (C.c_int, [_P, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), _PP]),
    "cugraph_amd_traversal_mg_plan_results": (C.c_int, [_P, _P, _P, _PP]),
    "cugraph_amd_traversal_mg_plan_keep_buffers": (None, [_P, C.c_int]),
    "cugraph_amd_traversal_mg_plan_rebind": (None, [_P, _P]),
    "cugraph_amd_traversal_mg_plan_free": (None, [_P]),
    "cugraph_amd_comm_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _PP, _PP]),
    "cugraph_amd_comm_free": (None, [_P]),
    "cugraph_amd_comm_host_selftest": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, _PP]),
    "cugraph_amd_comm_host_barrier": (C.c_int, [_P, _PP]),
    "cugraph_amd_comm_host_allgather": (C.c_int, [_P, _P, C.c_size_t, _P, _PP]),
    "cugraph_amd_comm_rank": (C.c_int, [_P]),
    "cugraph_amd_comm_size": (C.c_int, [_P]),
    "cugraph_amd_comm_selftest": (C.c_int, [_P, C.c_size_t, C.c_int, C.POINTER(C.c_double), _PP]),
    "cugraph_amd_read_matrix_market": (C.c_int, [_P, C.c_char_p, _PP, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int), _PP]),
    "cugraph_amd_handle_set_stream": (C.c_int, [_P, _P, _PP]),
    "cugraph_amd_handle_sync": (C.c_int, [_P, _PP]),
    "cugraph_amd_kernel_timing_enable": (None, [_P, C.c_int]),
    "cugraph_amd_kernel_timing_get": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_double), _PP]),
    "cugraph_amd_kernel_timing_reset": (None, [_P]),
    "cugraph_amd_kernel_timing_region_begin": (None, [_P, C.c_char_p]),
    "cugraph_amd_kernel_timing_region_end": (None, [_P, C.c_char_p]),
    "cugraph_amd_graph_num_vertices": (C.c_size_t, [_P]),
    "cugraph_amd_graph_num_edges": (C.c_size_t, [_P]),
    "cugraph_amd_graph_num_local_edges": (C.c_size_t, [_P]),
    "cugraph_amd_graph_compress_hypersparse": (C.c_int, [_P, _P, C.c_int, C.c_size_t, _PP]),
    "cugraph_amd_graph_hypersparse_view": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), _PP, _PP, _PP]),
    "cugraph_amd_set_pagerank_hot_tile": (C.c_int, [_P, C.c_int]),
    "cugraph_amd_last_traversal_stats": (None, [_P, C.POINTER(TraversalStats)]),
}

_lib = None


def lib() -> C.CDLL:
    """Loads libcugraph_c.so and attaches the prototypes.  Raises if the HIP library is absent."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension was not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C cugraph_amd/csrc`). cugraph_amd has no CPU fallback.")
        l = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(l, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib
