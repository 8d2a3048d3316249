"""TEST INFRASTRUCTURE: stand-in for the one function of python/pylibcugraph/pylibcugraph/utilities/api_tools.py that the modules on
the PageRank / BFS / SSSP path import (graphs.pyx:37): warn when the vertex / time columns of a graph do not share one dtype."""
import warnings


def ensure_valid_dtypes(src_or_offset_array, dst_or_index_array, vertices_array, edge_id_array, edge_start_time_array, edge_stop_time_array):
    vertex_types = {a.dtype for a in (src_or_offset_array, dst_or_index_array, vertices_array, edge_id_array) if a is not None}
    temporal_types = {a.dtype for a in (edge_start_time_array, edge_stop_time_array) if a is not None}
    if len(vertex_types) > 1:
        warnings.warn("The graph requires 'src_or_offset_array', 'dst_or_index_array' 'vertices_array' and 'edge_id_array' to match. "
                      "Those will be widened to 64-bit.", UserWarning)
    if len(temporal_types) > 1:
        warnings.warn("The graph requires 'edge_start_time_array' and 'edge_end_time_array' to match. Those will be widened to 64-bit.", UserWarning)
