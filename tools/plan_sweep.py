#!/usr/bin/env python3
"""GPU box: one RMAT graph, many PageRank plans -- each variant is a set of CUGRAPH_AMD_* environment settings read when the plan's
re-blocked structure is built (CUGRAPH_AMD_TILED_REBUILD forces a rebuild per plan).  Prints ms per iteration and the two kernel
averages per variant; variants are interleaved over REPS rounds so box drift shows up as spread, not as a ranking."""
import argparse
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=26)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--trim", action="store_true", help="hand the pool's cached blocks back to the driver before every plan build")
    ap.add_argument("variants", nargs="*", help='e.g. "base" "CUGRAPH_AMD_TP_TAIL_FRAC=0.1,CUGRAPH_AMD_TP_TAIL_CHUNK=2"')
    args = ap.parse_args()
    import torch

    import cugraph_amd as cg

    torch.cuda.set_device(0)
    h = cg.ResourceHandle()
    nv, ne = 1 << args.scale, 16 << args.scale
    src, dst = cg.generate_rmat_edgelist(h, args.scale, ne)
    g = cg.SGGraph(h, cg.GraphProperties(is_multigraph=True), src, dst, None, store_transposed=True, renumber=True,
                   vertices_array=torch.arange(nv, dtype=torch.int32, device="cuda"))
    del src, dst
    os.environ["CUGRAPH_AMD_TILED_REBUILD"] = "1"
    variants = args.variants or ["base"]
    for rep in range(args.reps):
        for v in variants:
            sets = [kv.split("=", 1) for kv in v.split(",") if "=" in kv]
            for k, val in sets:
                os.environ[k] = val
            if args.trim:
                from cugraph_amd import _capi as capi

                capi.lib().cugraph_amd_memory_pool_trim()
            t0 = time.perf_counter()
            plan = cg.PageRankPlan(h, g, 0.85)
            h.sync()
            plan_s = time.perf_counter() - t0
            plan.step(3)
            h.sync()
            h.kernel_timing(True)
            h.kernel_timing_reset()
            t0 = time.perf_counter()
            plan.step(args.steps)
            h.sync()
            dt = time.perf_counter() - t0
            n1, k1 = h.kernel_timing_get("pagerank_spmv")
            n2, k2 = h.kernel_timing_get("pagerank_reduce")
            h.kernel_timing(False)
            print(f"rep {rep} {v:60s} ms/iter {dt / args.steps * 1e3:.4f}  phase1 {k1 / max(n1, 1):.4f}  phase2 {k2 / max(n2, 1):.4f}  plan {plan_s:.2f} s", flush=True)
            del plan
            for k, _ in sets:
                os.environ.pop(k, None)


if __name__ == "__main__":
    main()
