"""Worker of the in-library multi-GPU tests (tests/test_mg_capi.py): one process per rank, all ranks on cuda:0 of the GPU box,
the library's own communicator (HIP IPC windows, peer writes) underneath -- no torch.distributed.
usage: ipc_worker.py <what> <session> <rank> <size> <outdir> [args...]"""
import json
import os
import sys
from pathlib import Path

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    what, session, rank, size, outdir = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), Path(sys.argv[5])
    args = sys.argv[6:]
    torch.cuda.set_device(0)
    import cugraph_amd as cg
    from cugraph_amd import _capi as capi
    from cugraph_amd.pylib import assert_success

    comm = cg.Comm(session, rank, size)
    h = cg.ResourceHandle(comm)
    assert h.rank == rank and h.comm_size == size
    out = {"rank": rank}
    if what == "selftest":
        n_words, iters = int(args[0]), int(args[1])
        res = (C.c_double * 4)()
        err = C.c_void_p()
        assert_success(capi.lib().cugraph_amd_comm_selftest(h.c_resource_handle_ptr, n_words, iters, res, C.byref(err)), err, "cugraph_amd_comm_selftest")
        out.update(barrier_us=res[0], push_gbps=res[1], multi_device=res[2])
    else:
        import mg_capi_cases

        out.update(mg_capi_cases.run(what, cg, h, comm, rank, size, outdir, args))
    (outdir / f"rank{rank}.json").write_text(json.dumps(out))
    del h
    comm.close()


if __name__ == "__main__":
    main()
