#!/usr/bin/env python3
"""GPU box: kernel variants that the library switches PER LAUNCH from the environment, timed alternately on ONE plan (same buffers, same placement: the
process-to-process spread of +-2.5 % does not enter).  usage: variant_ab.py SCALE "NAME=VALUE" [reps]  (compares unset against set); also checks that
two plans stepped with / without the variant hold bit-identical vectors."""
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

    scale = int(sys.argv[1])
    name, value = sys.argv[2].split("=", 1)
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    torch.cuda.set_device(0)
    h = cg.ResourceHandle()
    nv, ne = 1 << scale, 16 << scale
    src, dst = cg.generate_rmat_edgelist(h, scale, ne)
    g = cg.SGGraph(h, cg.GraphProperties(is_multigraph=True), src, dst, None, store_transposed=True, renumber=True,
                   vertices_array=torch.arange(nv, dtype=torch.int32, device="cuda"))
    del src, dst
    os.environ.pop(name, None)
    a = cg.PageRankPlan(h, g, 0.85)
    b = cg.PageRankPlan(h, g, 0.85)
    a.step(7)
    os.environ[name] = value.split(",")[0]
    b.step(7)
    os.environ.pop(name, None)
    _, pa, _ = a.result()
    _, pb, _ = b.result()
    print("bit-identical after 7 iterations:", bool(torch.equal(pa, pb)), flush=True)
    del b
    values = value.split(",")  # several values: every one of them against the default, on the same plan
    for rep in range(reps):
        for mode in ["base"] + [name + "=" + v for v in values]:
            if mode == "base":
                os.environ.pop(name, None)
            else:
                os.environ[name] = mode.split("=", 1)[1]
            print("rep %d %-28s ms/iter %.4f phase1 %.4f phase2 %.4f" % ((rep, mode) + timed(h, a, 20)), flush=True)
    os.environ.pop(name, None)


if __name__ == "__main__":
    main()
