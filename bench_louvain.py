#!/usr/bin/env python3
"""Louvain on a synthetic undirected RMAT graph (BASELINE.json config 5 names RMAT-26 on 8 GPUs -- this is
the single-GPU timing line for the row).

Graph: RMAT scale S, edge factor F, self-loops and duplicate pairs removed, both directions listed, integer weights 1..8 (so the
C oracle and the GPU must agree vertex for vertex).  Times cugraph_louvain end to end (all levels), reports edges * sweeps / s,
the modularity, and -- on a bounded sample -- the C oracle (oracle/oracle.c: orc_louvain, one core) beside it with a parity check.
Prints one JSON line.  Not the driver's bench (bench.py is).
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


def undirected_rmat(cg, h, scale, edge_factor, seed=5):
    """device tensors (src, dst, w) sorted by (src, dst); the same graph tests/test_gpu_parity.py: louvain_rmat_input builds"""
    import torch

    src, dst = cg.generate_rmat_edgelist(h, scale, edge_factor << scale, seed=seed)
    s, d = src.to(torch.int64), dst.to(torch.int64)
    keep = s != d
    lo, hi = torch.minimum(s[keep], d[keep]), torch.maximum(s[keep], d[keep])
    key = torch.unique(lo << 32 | hi)
    lo, hi = key >> 32, key & 0xFFFFFFFF
    wt = (1 + (lo * 7 + hi * 13) % 8).to(torch.float32)
    s2, d2, w2 = torch.cat([lo, hi]), torch.cat([hi, lo]), torch.cat([wt, wt])
    order = torch.argsort(s2 << 32 | d2)
    return s2[order].to(torch.int32), d2[order].to(torch.int32), w2[order]


def louvain_bench(cg, h, scale=22, edge_factor=8, repeats=3, cpu_scale=18):
    """Times cugraph_louvain on the undirected RMAT graph of `scale`; returns the dict of the JSON line (roofline, cpu_baseline, check)."""
    import numpy as np
    import torch

    def run(sc, reps):
        nv = 1 << sc
        src, dst, w = undirected_rmat(cg, h, sc, edge_factor)
        g = cg.SGGraph(h, cg.GraphProperties(is_symmetric=True), src, dst, w, renumber=False, vertices_array=torch.arange(nv, dtype=torch.int32, device="cuda"))
        times = []
        for _ in range(reps + 1):  # one warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            v, c, q = cg.louvain(h, g, 100, 1e-7, 1.0, False)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        return src, dst, w, v, c, q, times[1:], h.last_traversal_stats()

    src, dst, w, v, c, q, times, work = run(scale, repeats)
    ne = int(src.numel())
    best = min(times)
    # Algorithmic bytes: a local-moving sweep over a level with E_l directed edges and V_l vertices must read every
    # edge's (destination, weight) and the cluster of its destination (4 + 8 + 4 bytes) and, per vertex, its offset, cluster, weight and
    # the weight of its cluster, and write its new cluster (4 + 4 + 8 + 8 + 4 bytes); a contraction reads and writes the level's edges once
    # (2 x 16 bytes per edge).  Sums over the sweeps / levels come from the library (cugraph_amd_last_traversal_stats).
    alg = 16 * work["edges_inspected"] + 28 * work["vertices_reached"] + 32 * work["edges_of_reached"]
    from bench_traversal import counter_traffic

    traffic, tsrc = counter_traffic(f"louvain_s{scale}")
    out = {
        "metric": f"louvain_seconds_rmat{scale}", "value": round(best, 4), "unit": "s", "higher_is_better": False, "n_gpus": 1,
        "config": {"workload": f"Louvain (max_level 100, threshold 1e-7, resolution 1), undirected simple RMAT scale {scale} edge factor {edge_factor}, "
                               f"integer weights 1..8, both directions stored", "vertices": 1 << scale, "directed_edges": ne},
        "modularity": q, "clusters": int(torch.unique(c).numel()), "seconds_all": [round(t, 4) for t in times], "sweeps": int(work["steps"]),
        "directed_edges_per_second": round(ne / best, 1), "edge_sweeps_per_second": round(work["edges_inspected"] / best, 1), "dtype": "f64", "data": "synthetic",
        "roofline": {"bound": "hbm", "achieved": round(alg / best / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(alg / best / 1e9 / 8000.0, 4),
                     "traffic": traffic, "traffic_source": tsrc, "algorithmic_bytes": int(alg),
                     "kernel": "whole call (all levels: sort + segment passes + contraction); 16 B x edge-sweeps + 28 B x vertex-sweeps + 32 B x contracted edges"},
    }
    # the returned clustering's modularity, recomputed from the edge list with torch (fp64), must be the reported one
    # (prefix sums over the source-sorted edge list and over the cluster-sorted vertices: fp64 atomics -- index_add_ -- take seconds here)
    nvv = 1 << scale
    cl = torch.empty(nvv, dtype=torch.int64, device="cuda")
    cl[v.to(torch.int64)] = c.to(torch.int64)
    wd = w.double()
    m = wd.sum()
    zero = torch.zeros(1, dtype=torch.float64, device="cuda")
    c0 = torch.cat([zero, torch.cumsum(wd, 0)])
    off = torch.searchsorted(src.long().contiguous(), torch.arange(nvv + 1, device="cuda"))
    k = c0[off[1:]] - c0[off[:-1]]                      # vertex weights (edges are sorted by source)
    order = torch.argsort(cl)
    _, counts = torch.unique_consecutive(cl[order], return_counts=True)
    ends = torch.cumsum(counts, 0)
    c1 = torch.cat([zero, torch.cumsum(k[order], 0)])
    a = c1[ends] - c1[ends - counts]                     # cluster weights
    internal = wd[cl[src.long()] == cl[dst.long()]].sum()
    q_torch = float(internal / m - (a * a).sum() / (m * m))
    out["check"] = {"modularity_recomputed_abs_err": abs(q_torch - q), "ok": abs(q_torch - q) <= 1e-9,
                    "what": "modularity of the returned clustering recomputed from the edge list (torch, fp64)"}
    if cpu_scale:
        from oracle import oracle as orc

        s2, d2, w2, v2, c2, q2, t2, _ = (src, dst, w, v, c, q, times, work) if cpu_scale == scale else run(cpu_scale, 1)
        s_h, d_h, w_h = s2.cpu().numpy(), d2.cpu().numpy(), w2.cpu().numpy()
        t0 = time.perf_counter()
        oc, oq, olevels, osweeps = orc.louvain_c(1 << cpu_scale, s_h, d_h, w_h, 100, 1e-7, 1.0)
        cpu_s = time.perf_counter() - t0
        got = np.empty(1 << cpu_scale, np.int64)
        got[v2.cpu().numpy()] = c2.cpu().numpy()
        out["cpu_baseline"] = {"value": round(cpu_s, 3), "unit": "s", "cores": 1, "kind": "port",
                               "sample": f"oracle/oracle.c orc_louvain, RMAT-{cpu_scale} (same construction), {olevels} levels, {osweeps} sweeps",
                               "gpu_seconds_same_graph": round(min(t2), 4)}
        out["check"].update({"clusters_equal_oracle": bool(np.array_equal(got, oc)), "modularity_abs_err_vs_oracle": abs(q2 - oq), "oracle_scale": cpu_scale})
        out["check"]["ok"] = bool(out["check"]["ok"] and np.array_equal(got, oc) and abs(q2 - oq) <= 1e-9)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=22)
    ap.add_argument("--edge-factor", type=int, default=8)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--cpu-scale", type=int, default=18, help="RMAT scale of the bounded CPU sample (0 = skip)")
    ap.add_argument("--out", type=str, default=None)
    args = ap.parse_args()

    import torch

    import cugraph_amd as cg

    torch.cuda.set_device(0)
    h = cg.ResourceHandle()
    out = louvain_bench(cg, h, args.scale, args.edge_factor, args.repeats, args.cpu_scale)
    line = json.dumps(out)
    print(line, flush=True)
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(line + "\n")


if __name__ == "__main__":
    main()
