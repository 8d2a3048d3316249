"""Worker for the multi-process BFS / SSSP tests (launched by tests/test_mg_traversal.py, one process per rank).
engine "numpy": gloo on CPU, local compute = tests/numpy_traversal_engine.py (exercises partitioning + collectives);
engine "hip":   gloo, every rank drives the HIP engine on cuda:0 (exercises the partitioned HIP kernels on one GPU)."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from cugraph_amd import mg_traversal as mt  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from numpy_traversal_engine import NumpyTraversalEngine  # noqa: E402


def main():
    engine, algo, scale, out_dir = sys.argv[1], sys.argv[2], int(sys.argv[3]), Path(sys.argv[4])
    sources = [int(x) for x in sys.argv[5].split(",")]
    limit = float(sys.argv[6])  # BFS: depth limit (< 0: none); SSSP: cutoff (< 0: none)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nv, ne = 1 << scale, 16 << scale
    per = (ne + world - 1) // world
    s, d = orc.rmat(scale, min(per, ne - rank * per), first_edge=rank * per)
    factory = NumpyTraversalEngine if engine == "numpy" else None
    if factory is None:
        torch.cuda.set_device(0)
    levels = bu_levels = 0
    if algo == "bfs":
        t = mt.MGTraversal(torch.from_numpy(s), torch.from_numpy(d), nv, None, "bfs", None, factory)
        v, dd, pp = t.run(sources, depth_limit=(None if limit < 0 else int(limit)))
        levels, bu_levels = t.levels, t.bottom_up_levels
    else:
        w = np.random.default_rng(1).integers(1, 256, size=ne).astype(np.float32)[rank * per: rank * per + s.size].copy()
        v, dd, pp = mt.sssp(torch.from_numpy(s), torch.from_numpy(d), torch.from_numpy(w), nv, sources[0],
                            cutoff=(mt.FLT_MAX if limit < 0 else limit), engine_factory=factory)
    np.savez(out_dir / f"rank{rank}.npz", v=v.cpu().numpy(), d=dd.cpu().numpy(), p=pp.cpu().numpy(), levels=levels, bottom_up_levels=bu_levels)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
