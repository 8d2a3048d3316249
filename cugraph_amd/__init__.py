"""cugraph_amd -- MI355X-native PageRank / BFS / SSSP core behind the cuGraph C API.

Layout (only what the hot path needs):
  csrc/      HIP kernels + the C-ABI implementation (-> lib/libcugraph_c.so)
  _capi.py   ctypes binding of that C ABI
  pylib.py   host-side mirror of the reference's pylibcugraph interface for this path
  csrc/comm.hpp  the library's own one-node communicator (HIP IPC windows + peer writes over xGMI): MG behind the C API
  mg.py      multi-GPU PageRank (one process per GPU, torch.distributed / RCCL)
  mg_traversal.py  multi-GPU BFS / SSSP (same process model)
"""
from .pylib import (  # noqa: F401
    Comm,
    FailedToConvergeError,
    GraphProperties,
    MGGraph,
    PageRankPlan,
    ResourceHandle,
    SGGraph,
    bfs,
    bfs_extract_paths,
    decompress_to_edgelist,
    degrees,
    generate_rmat_edgelist,
    has_vertex,
    in_degrees,
    louvain,
    out_degrees,
    pagerank,
    personalized_pagerank,
    read_matrix_market,
    rmat_edgelist,
    sssp,
)

__version__ = "0.1.0"
