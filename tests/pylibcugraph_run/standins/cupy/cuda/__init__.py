"""TEST INFRASTRUCTURE: cupy.cuda.UnownedMemory / MemoryPointer as utils.pyx:279-283 uses them (a view of library-owned memory)."""
import ctypes

import torch


class UnownedMemory:
    def __init__(self, ptr, size, owner, device_id=-1):
        self.ptr, self.size, self.owner = int(ptr), int(size), owner


class MemoryPointer:
    def __init__(self, mem, offset):
        self.mem, self.offset = mem, int(offset)

    @property
    def ptr(self):
        return self.mem.ptr + self.offset


def _copy_from_pointer(memptr, n, tdtype):
    out = torch.empty(n, dtype=tdtype, device="cuda")
    if n:
        hip = ctypes.CDLL("libamdhip64.so")
        rc = hip.hipMemcpy(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(memptr.ptr), ctypes.c_size_t(n * out.element_size()), 3)  # device to device
        if rc != 0:
            raise RuntimeError(f"hipMemcpy failed: {rc}")
    return out
