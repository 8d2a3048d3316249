"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol include/*.h declares.
No compute call is made (there is no GPU in the build container)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    names = set()
    for h in list((ROOT / "include" / "cugraph_c").glob("*.h")) + list((ROOT / "include" / "cugraph_amd").glob("*.h")):
        text = h.read_text()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"CUGRAPH_EXPORT\s+[^;{]*?\b(cugraph_[a-z0-9_]+)\s*\(", text):
            names.add(m.group(1))
    return names


@pytest.fixture(scope="module")
def library():
    from cugraph_amd import _capi

    if not _capi.LIB_PATH.exists():
        _capi.build()
    return _capi


def test_headers_declare_the_boundary():
    syms = declared_symbols()
    for must in ("cugraph_create_resource_handle", "cugraph_graph_create_with_times_sg", "cugraph_pagerank_allow_nonconvergence",
                 "cugraph_personalized_pagerank", "cugraph_bfs", "cugraph_sssp", "cugraph_has_vertex",
                 "cugraph_type_erased_device_array_view_copy", "cugraph_centrality_result_get_values",
                 "cugraph_paths_result_get_predecessors", "cugraph_error_message"):
        assert must in syms
    assert len(syms) >= 60


def test_library_exports_every_declared_symbol(library):
    lib = ctypes.CDLL(str(library.LIB_PATH))
    missing = [s for s in sorted(declared_symbols()) if not hasattr(lib, s)]
    assert not missing, missing


def test_ctypes_prototypes_cover_the_headers(library):
    assert set(library.PROTOTYPES) == declared_symbols()
    library.lib()  # attaches every prototype; AttributeError if one is absent


def test_sonames_of_the_reference_wheel(library):
    # python/libcugraph/libcugraph/load.py:57 dlopens these three
    for n in ("libcugraph.so", "libcugraph_mg.so", "libcugraph_c.so"):
        assert (library.LIB_DIR / n).exists()


def test_no_gpu_means_loud_failure_not_fallback(library):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    l = library.lib()
    assert l.cugraph_amd_version().startswith(b"cugraph_amd")
    assert l.cugraph_create_resource_handle(None) is None  # NULL handle, as the header documents
    from cugraph_amd import ResourceHandle

    with pytest.raises(RuntimeError):
        ResourceHandle()


def test_error_objects_are_null_safe(library):
    l = library.lib()
    l.cugraph_error_free(None)
    assert l.cugraph_error_message(None) is None
    assert l.cugraph_resource_handle_get_comm_size(None) == 0
    # host arrays work without a device
    arr, err = ctypes.c_void_p(), ctypes.c_void_p()
    assert l.cugraph_type_erased_host_array_create(None, 5, library.INT32, ctypes.byref(arr), ctypes.byref(err)) == 0
    v = l.cugraph_type_erased_host_array_view(arr)
    assert l.cugraph_type_erased_host_array_size(v) == 5 and l.cugraph_type_erased_host_array_type(v) == library.INT32
    l.cugraph_type_erased_host_array_view_free(v)
    l.cugraph_type_erased_host_array_free(arr)
    # array.h:207: release hands the malloc'ed storage to the caller; the array object survives as an empty array
    assert l.cugraph_type_erased_host_array_create(None, 3, library.INT64, ctypes.byref(arr), ctypes.byref(err)) == 0
    v = l.cugraph_type_erased_host_array_view(arr)
    ptr = l.cugraph_type_erased_host_array_pointer(v)
    l.cugraph_type_erased_host_array_view_free(v)
    ctypes.memmove(ptr, (ctypes.c_int64 * 3)(7, 8, 9), 24)
    raw = l.cugraph_type_erased_host_array_release(arr)
    assert raw == ptr and list((ctypes.c_int64 * 3).from_address(raw)) == [7, 8, 9]
    v = l.cugraph_type_erased_host_array_view(arr)
    assert l.cugraph_type_erased_host_array_size(v) == 0
    l.cugraph_type_erased_host_array_view_free(v)
    l.cugraph_type_erased_host_array_free(arr)  # must not free `raw`
    assert list((ctypes.c_int64 * 3).from_address(raw)) == [7, 8, 9]
    ctypes.CDLL(None).free(ctypes.c_void_p(raw))
    assert l.cugraph_type_erased_host_array_release(None) is None and l.cugraph_type_erased_device_array_release(None) is None
    # a NULL handle is reported, not dereferenced
    code = l.cugraph_type_erased_device_array_create(None, 4, library.INT32, ctypes.byref(arr), ctypes.byref(err))
    assert code == library.CUGRAPH_INVALID_HANDLE
    assert b"NULL" in l.cugraph_error_message(err)
    l.cugraph_error_free(err)
