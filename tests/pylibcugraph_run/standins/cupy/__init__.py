"""TEST INFRASTRUCTURE: the sliver of cupy that pylibcugraph's Cython modules touch on the PageRank / BFS / SSSP path, on top of
torch device tensors (cupy is not installed in this image; SURVEY.md section 7 "hard parts").  Used by: utils.pyx:175-196
(cupy.zeros + __cuda_array_interface__ for result columns), utils.pyx:268-288 (ndarray / cuda.UnownedMemory / MemoryPointer
views of library-owned memory), sssp.pyx:124 (cupy.asarray of the source)."""
import numpy
import torch

from . import cuda  # noqa: F401

_NP2T = {numpy.dtype("int8"): torch.int8, numpy.dtype("int16"): torch.int16, numpy.dtype("int32"): torch.int32, numpy.dtype("int64"): torch.int64,
         numpy.dtype("uint8"): torch.uint8, numpy.dtype("float32"): torch.float32, numpy.dtype("float64"): torch.float64, numpy.dtype("bool"): torch.bool}


class ndarray:
    """1-D device array: a torch tensor on cuda:0 with numpy dtype semantics."""

    def __init__(self, shape=0, dtype=numpy.float32, memptr=None, tensor=None):
        self.dtype = numpy.dtype(dtype)
        if tensor is not None:
            self._t = tensor
        elif memptr is not None:  # view of foreign memory: copy it out while it is alive (the views pylibcugraph builds are read once)
            n = int(shape if isinstance(shape, int) else shape[0])
            self._t = cuda._copy_from_pointer(memptr, n, _NP2T[self.dtype])
        else:
            n = int(shape if isinstance(shape, int) else shape[0])
            self._t = torch.zeros(n, dtype=_NP2T[self.dtype], device="cuda")

    @property
    def __cuda_array_interface__(self):
        return {"shape": (self._t.numel(),), "typestr": self.dtype.str, "data": (self._t.data_ptr() if self._t.numel() else 0, False), "version": 3, "strides": None}

    @property
    def shape(self):
        return (self._t.numel(),)

    @property
    def size(self):
        return self._t.numel()

    def __len__(self):
        return self._t.numel()

    def get(self):
        return self._t.cpu().numpy()

    def tolist(self):
        return self._t.cpu().tolist()

    def __getitem__(self, i):
        r = self._t[i]
        return r.item() if r.dim() == 0 else ndarray(tensor=r.contiguous(), dtype=self.dtype)


def zeros(shape, dtype=numpy.float32):
    return ndarray(shape, dtype)


def asarray(obj, dtype=None):
    if isinstance(obj, ndarray):
        return obj if dtype is None or numpy.dtype(dtype) == obj.dtype else ndarray(tensor=obj._t.to(_NP2T[numpy.dtype(dtype)]), dtype=dtype)
    a = numpy.asarray(obj, dtype=dtype)
    return ndarray(tensor=torch.as_tensor(numpy.ascontiguousarray(a), device="cuda"), dtype=a.dtype)


array = asarray


def asnumpy(a):
    return a.get() if isinstance(a, ndarray) else numpy.asarray(a)
