#!/usr/bin/env python3
"""BFS / SSSP throughput on synthetic RMAT (BASELINE.json config 3: RMAT-24 BFS + SSSP on one MI355X).

Graph500-style protocol (cpp/tests/traversal/mg_graph500_bfs_test.cu:113-114, 757-763): N roots with non-zero
out-degree chosen by a fixed-seed hash, one warm-up, TEPS = out-edges of the reached vertices / time, harmonic mean.
SSSP weights: all 1.0f ("integer hops": distances must equal the BFS distances bit for bit) or integers 1..255.
Prints one JSON line.  Not the driver's bench (bench.py is); kept beside it because BASELINE.json names these configs.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    import numpy as np
    import torch

    import cugraph_amd as cg

    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=24)
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--roots", type=int, default=16)
    ap.add_argument("--weights", choices=["unit", "int"], default="unit")
    ap.add_argument("--symmetric", action="store_true", help="add the reverse of every edge (Graph500 input)")
    ap.add_argument("--no-sssp", action="store_true")
    ap.add_argument("--predecessors", action="store_true")
    args = ap.parse_args()

    torch.cuda.set_device(0)
    h = cg.ResourceHandle()
    nv, ne = 1 << args.scale, args.edge_factor << args.scale
    src, dst = cg.generate_rmat_edgelist(h, args.scale, ne)
    if args.symmetric:
        src, dst = torch.cat([src, dst]), torch.cat([dst, src])
        ne *= 2
    if args.weights == "unit":
        w = torch.ones(ne, dtype=torch.float32, device="cuda")
    else:
        g_ = torch.Generator(device="cuda").manual_seed(1)
        w = torch.randint(1, 256, (ne,), generator=g_, device="cuda").to(torch.float32)
    verts = torch.arange(nv, dtype=torch.int32, device="cuda")
    t0 = time.perf_counter()
    g = cg.SGGraph(h, cg.GraphProperties(is_multigraph=True, is_symmetric=args.symmetric), src, dst, w, store_transposed=False, renumber=True,
                   vertices_array=verts)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    outdeg = torch.bincount(src.to(torch.int64), minlength=nv)
    del src, dst
    # roots: fixed-seed hash order over the vertices with out-edges
    cand = torch.nonzero(outdeg > 0).flatten()
    perm = torch.randperm(cand.numel(), generator=torch.Generator().manual_seed(0))[: args.roots]
    roots = cand[perm.to(cand.device)].to(torch.int32)

    def run(kind):
        times, edges, steps = [], [], []
        for i, r in enumerate([roots[0], roots[0]] + list(roots)):  # two warm-ups (the second BFS of a directed graph builds its CSC)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if kind == "bfs":
                d, p, v = cg.bfs(h, g, r.reshape(1).clone(), False, 0, args.predecessors, False)
            else:
                v, d, p = cg.sssp(h, g, int(r), 3.0e38, args.predecessors, False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            st = h.last_traversal_stats()
            if i > 1:
                times.append(dt)
                edges.append(st["edges_of_reached"] if kind == "bfs" else None)
                steps.append(st["steps"])
            last = (v, d)
        return times, edges, steps, last

    out = {"workload": f"RMAT scale {args.scale} edge factor {args.edge_factor}{' symmetrised' if args.symmetric else ''}, {args.roots} roots, "
                       f"weights {args.weights}", "vertices": nv, "edges": ne, "graph_build_s": round(build_s, 3)}
    bt, be, bs, (bv, bd) = run("bfs")
    teps = [e / t for e, t in zip(be, bt)]
    out["bfs"] = {"mean_ms": round(1e3 * float(np.mean(bt)), 3), "min_ms": round(1e3 * float(np.min(bt)), 3), "max_ms": round(1e3 * float(np.max(bt)), 3), "harmonic_mean_mteps": round(len(teps) / sum(1.0 / x for x in teps) / 1e6, 1),
                  "mean_levels": float(np.mean(bs)), "mean_edges_of_reached": float(np.mean(be))}
    if not args.no_sssp:
        st, _, ss, (sv, sd) = run("sssp")
        teps = [e / t for e, t in zip(be, st)]  # same roots: same reached set, scored on the same edge count
        out["sssp"] = {"mean_ms": round(1e3 * float(np.mean(st)), 3), "min_ms": round(1e3 * float(np.min(st)), 3), "max_ms": round(1e3 * float(np.max(st)), 3), "harmonic_mean_mteps": round(len(teps) / sum(1.0 / x for x in teps) / 1e6, 1),
                       "mean_steps": float(np.mean(ss))}
        if args.weights == "unit":  # integer hops: bit-exact against BFS (last root)
            a = torch.empty(nv, dtype=torch.int64, device="cuda"); a[bv.to(torch.int64)] = bd.to(torch.int64)
            b = torch.empty(nv, dtype=torch.float32, device="cuda"); b[sv.to(torch.int64)] = sd
            reach = a != 2147483647
            ok = bool(torch.equal(a[reach].to(torch.float32), b[reach])) and bool((b[~reach] == torch.finfo(torch.float32).max).all())
            out["sssp"]["unit_weight_distances_equal_bfs"] = ok
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
