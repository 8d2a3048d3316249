"""One-rank check of the 1-D multi-GPU PageRank against the 2-D one and the single-GPU library at a given RMAT scale
(GPU box; `python tools/debug_mg1d.py 22 24`).  Also checks the library primitives of the plan build against torch's."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cugraph_amd import mg  # noqa: E402
from cugraph_amd.pylib import ResourceHandle, generate_rmat_edgelist  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
h = ResourceHandle()
for scale in [int(a) for a in sys.argv[1:]] or [22, 24]:
    nv, ne = 1 << scale, 16 << scale
    src, dst = generate_rmat_edgelist(h, scale, ne, first_edge=0)
    out = {}
    layouts = os.environ.get("DBG_LAYOUTS", "1d,2d,1d,2d").split(",")
    for name in layouts:
        cls = mg.MGPageRank if name == "1d" else mg.MGPageRank2D
        pr = cls(src, dst, nv, alpha=0.85)
        masses = []
        for _ in range(6):
            pr.step(1)
            if os.environ.get("DBG_DEVSYNC") == "1":
                torch.cuda.synchronize()
            torch.cuda.synchronize()
            masses.append(float(pr.result()[1].double().sum()))
        verts, vals = pr.result()
        full = torch.zeros(nv, dtype=torch.float64, device=vals.device)
        full[verts] = vals.double()
        out[name] = full
        print(scale, name, "mass after 1..6 iterations", [round(m, 6) for m in masses], flush=True)
        if name == "1d":
            ex = pr.ex
            print(scale, "ncols", ex.ncols, "recv", ex.recv_counts, "send", ex.send_counts, flush=True)
        del pr
    if len(out) < 2:
        continue
    d = (out["1d"] - out["2d"]).abs()
    print(scale, "max |1d - 2d|", float(d.max()), "rows differing > 1e-9", int((d > 1e-9).sum()), "sum diff", float((out["1d"] - out["2d"]).sum()), flush=True)
    bad = torch.nonzero(d > 1e-9).flatten()[:10]
    print(scale, "first differing vertices", bad.tolist(), out["1d"][bad].tolist(), out["2d"][bad].tolist(), flush=True)
    del out, src, dst
dist.destroy_process_group()
