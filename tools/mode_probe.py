#!/usr/bin/env python3
"""GPU box: is the +-4 % spread of the PageRank iteration a property of the PROCESS (placement of the plan's buffers) or of the moment (clocks)?
One process, one graph: the same plan timed several times in a row (moment), then plans rebuilt after the pool has been trimmed and a dummy
allocation of varying size has moved the physical placement (placement)."""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def timed(h, plan, steps):
    plan.step(3)
    h.sync()
    h.kernel_timing(True)
    h.kernel_timing_reset()
    t0 = time.perf_counter()
    plan.step(steps)
    h.sync()
    dt = time.perf_counter() - t0
    n1, k1 = h.kernel_timing_get("pagerank_spmv")
    n2, k2 = h.kernel_timing_get("pagerank_reduce")
    h.kernel_timing(False)
    return dt / steps * 1e3, k1 / max(n1, 1), k2 / max(n2, 1)


def main():
    import torch

    import cugraph_amd as cg
    from cugraph_amd import _capi as capi

    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    torch.cuda.set_device(0)
    h = cg.ResourceHandle()
    nv, ne = 1 << scale, 16 << scale
    src, dst = cg.generate_rmat_edgelist(h, scale, ne)
    g = cg.SGGraph(h, cg.GraphProperties(is_multigraph=True), src, dst, None, store_transposed=True, renumber=True,
                   vertices_array=torch.arange(nv, dtype=torch.int32, device="cuda"))
    del src, dst
    os.environ["CUGRAPH_AMD_TILED_REBUILD"] = "1"
    plan = cg.PageRankPlan(h, g, 0.85)
    for k in range(5):
        print("same plan   %d: ms/iter %.4f phase1 %.4f phase2 %.4f" % ((k,) + timed(h, plan, 20)), flush=True)
    time.sleep(2.0)
    print("after 2 s idle: ms/iter %.4f phase1 %.4f phase2 %.4f" % timed(h, plan, 20), flush=True)
    print("200 steps     : ms/iter %.4f phase1 %.4f phase2 %.4f" % timed(h, plan, 200), flush=True)
    del plan
    dummies = []
    for k in range(6):
        capi.lib().cugraph_amd_memory_pool_trim()
        torch.cuda.empty_cache()
        dummies.append(torch.empty((k * 37 + 5) << 20, dtype=torch.uint8, device="cuda"))  # 5, 42, 79 ... MiB stay allocated
        plan = cg.PageRankPlan(h, g, 0.85)
        print("rebuilt plan %d: ms/iter %.4f phase1 %.4f phase2 %.4f" % ((k,) + timed(h, plan, 20)), flush=True)
        del plan


if __name__ == "__main__":
    main()
