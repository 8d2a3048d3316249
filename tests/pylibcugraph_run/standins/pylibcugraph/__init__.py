"""TEST INFRASTRUCTURE.  Package shell around the reference's pylibcugraph Cython modules, which tests/pylibcugraph_run/build.sh
compiles UNCHANGED and in place from the reference tree against include/ and links to cugraph_amd/lib/libcugraph_c.so.
The reference's own __init__.py imports all ~70 algorithm modules; this one imports the ones on the PageRank / BFS / SSSP
path (python/pylibcugraph/pylibcugraph/__init__.py:15-158 lists them under the same names)."""
from pylibcugraph.graphs import SGGraph, MGGraph  # noqa: F401
from pylibcugraph.resource_handle import ResourceHandle  # noqa: F401
from pylibcugraph.graph_properties import GraphProperties  # noqa: F401
from pylibcugraph.pagerank import pagerank  # noqa: F401
from pylibcugraph.personalized_pagerank import personalized_pagerank  # noqa: F401
from pylibcugraph.sssp import sssp  # noqa: F401
from pylibcugraph.bfs import bfs  # noqa: F401
from pylibcugraph.louvain import louvain  # noqa: F401
from pylibcugraph.random import CuGraphRandomState  # noqa: F401
from pylibcugraph.generate_rmat_edgelist import generate_rmat_edgelist  # noqa: F401
from pylibcugraph.degrees import in_degrees, out_degrees, degrees  # noqa: F401
from pylibcugraph.decompress_to_edgelist import decompress_to_edgelist  # noqa: F401
from pylibcugraph.has_vertex import has_vertex  # noqa: F401
from pylibcugraph import exceptions  # noqa: F401
