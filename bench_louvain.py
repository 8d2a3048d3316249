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
    if os.environ.get("BENCH_LOUVAIN_RELABEL") == "degree":  # experiment: the same graph with its vertices numbered by descending degree (another clustering: ties break on ids)
        deg = torch.bincount(s2, minlength=1 << scale)
        rank = torch.empty_like(deg)
        rank[torch.argsort(-deg, stable=True)] = torch.arange(deg.numel(), device=deg.device)
        s2, d2 = rank[s2], rank[d2]
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
                     "kernel": "whole call (all levels: evaluation sweeps over LDS tables, moves, contraction by radix sort); 16 B x edge-sweeps + 28 B x vertex-sweeps + 32 B x contracted edges"},
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


def louvain_bench_mg(args):
    """BASELINE config 5 (Louvain partitioned over the GPUs of one node) on the library's communicator: every rank builds a slice of the same
    undirected RMAT graph, cugraph_graph_create_mg + cugraph_louvain (collective).  Launched under torch.distributed.run (RANK / WORLD_SIZE /
    LOCAL_RANK / MASTER_PORT from the environment) -- `python bench_louvain.py --gpus N` starts the ranks itself.  The clustering is checked
    against the committed fixture of the C oracle when there is one for the scale (sha256 of the assembled cluster column)."""
    import hashlib

    import numpy as np
    import torch

    import cugraph_amd as cg

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    single = os.environ.get("CUGRAPH_AMD_MG_TEST_SINGLE_GPU") == "1"  # all ranks share cuda:0 (the IPC path is the same)
    torch.cuda.set_device(0 if single else int(os.environ.get("LOCAL_RANK", str(rank))))
    comm = cg.Comm(f"louvain_{os.environ.get('MASTER_PORT', '0')}", rank, world)
    h = cg.ResourceHandle(comm)
    scale, nv = args.scale, 1 << args.scale
    src, dst, w = undirected_rmat(cg, h, scale, args.edge_factor)  # (every rank generates the list and keeps a slice: generation is not timed)
    ne = int(src.numel())
    mine = torch.arange(ne, device="cuda") % world == rank
    s_m, d_m, w_m = src[mine].contiguous(), dst[mine].contiguous(), w[mine].contiguous()
    del src, dst, w, mine
    verts = torch.arange(rank, nv, world, dtype=torch.int32, device="cuda")
    t0 = time.perf_counter()
    g = cg.MGGraph(h, cg.GraphProperties(is_symmetric=True), [s_m], [d_m], [w_m], store_transposed=False, vertices_array=[verts])
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    times = []
    for _ in range(args.repeats + 1):  # one warm-up
        h.sync()
        comm.barrier()
        t0 = time.perf_counter()
        v, c, q = cg.louvain(h, g, 100, 1e-7, 1.0, False)
        h.sync()
        comm.barrier()
        times.append(max(x[0] for x in comm.allgather_f64([time.perf_counter() - t0])))
    work = h.last_traversal_stats()
    best = min(times[1:])
    # the whole job's algorithmic bytes: edge-sweeps summed over the ranks (each rank sweeps its share), vertex-sweeps once
    tot = comm.allgather_f64([float(work["edges_inspected"]), float(work["edges_of_reached"])])
    e_sweeps, e_contr = sum(t[0] for t in tot), sum(t[1] for t in tot)
    alg = 16 * e_sweeps + 28 * work["vertices_reached"] + 32 * e_contr
    # assemble the cluster column on rank 0 through files (checker only) and compare with the fixture
    tmp = Path(os.environ.get("TMPDIR", "/tmp")) / f"louvain_mg_{os.environ.get('MASTER_PORT', '0')}"
    tmp.mkdir(parents=True, exist_ok=True)
    np.savez(tmp / f"rank{rank}.npz", v=v.cpu().numpy(), c=c.cpu().numpy())
    comm.barrier()
    out = None
    if rank == 0:
        col = np.full(nv, -1, np.int64)
        for r in range(world):
            z = np.load(tmp / f"rank{r}.npz")
            col[z["v"]] = z["c"]
        check = {"every_vertex_once": bool((col >= 0).all()), "ok": bool((col >= 0).all())}
        fx = ROOT / "tests" / "golden" / f"louvain_rmat{scale}.json"
        if fx.exists() and args.edge_factor == 8:
            gold = json.loads(fx.read_text())
            same = hashlib.sha256(np.ascontiguousarray(col, np.int32).tobytes()).hexdigest() == gold["clusters_sha256"]
            check.update({"clusters_equal_oracle_fixture": bool(same), "modularity_abs_err_vs_fixture": abs(q - gold["modularity"]), "fixture": fx.name})
            check["ok"] = bool(check["ok"] and same and abs(q - gold["modularity"]) <= 1e-9)
        out = {
            "metric": f"louvain_seconds_rmat{scale}", "value": round(best, 4), "unit": "s", "higher_is_better": False, "n_gpus": world, "scaling": "strong",
            "config": {"workload": f"Louvain (max_level 100, threshold 1e-7, resolution 1), undirected simple RMAT scale {scale} edge factor {args.edge_factor}, integer weights 1..8, "
                                   "both directions stored; cugraph_graph_create_mg + cugraph_louvain on the library's communicator: cyclic vertex ownership, label merge by peer "
                                   "pushes per sweep, two-stage integer contraction", "vertices": nv, "directed_edges": ne, "parallelism": f"{world} ranks, 1 process per rank",
                       "all_ranks_on_one_gpu": single},
            "modularity": q, "clusters": int(np.unique(col).size), "seconds_all": [round(t, 4) for t in times[1:]], "sweeps": int(work["steps"]), "graph_build_s": round(build_s, 3),
            "dtype": "f64", "data": "synthetic",
            "roofline": {"bound": "hbm", "achieved": round(alg / best / 1e9, 1), "peak": 8000.0 * (1 if single else world), "unit": "GB/s",
                         "frac": round(alg / best / 1e9 / (8000.0 * (1 if single else world)), 4), "traffic": None, "algorithmic_bytes": int(alg),
                         "kernel": "whole call on all ranks; 16 B x edge-sweeps + 28 B x vertex-sweeps + 32 B x contracted edges"},
            "check": check,
        }
    comm.barrier()
    del g
    h.sync()
    comm.barrier()
    del h
    comm.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="> 1: the partitioned run on the library's communicator (one process per GPU)")
    ap.add_argument("--scale", type=int, default=22)
    ap.add_argument("--edge-factor", type=int, default=8)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--cpu-scale", type=int, default=18, help="RMAT scale of the bounded CPU sample (0 = skip)")
    ap.add_argument("--out", type=str, default=None)
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:  # start the ranks ourselves, as bench.py does
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               str(Path(__file__).resolve())] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ)))
    if args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        out = louvain_bench_mg(args)
        if out is not None:
            line = json.dumps(out)
            print(line, flush=True)
            if args.out:
                Path(args.out).parent.mkdir(parents=True, exist_ok=True)
                Path(args.out).write_text(line + "\n")
        return

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
