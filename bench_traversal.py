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


def main_partitioned_ipc(args):
    """Partitioned BFS / SSSP through the reference's own entry points on the library's communicator (cugraph_graph_create_mg + cugraph_bfs /
    cugraph_sssp; csrc/traversal_mg_driver.hip): the SAME RMAT graph over all ranks (strong scaling).  Launched under torch.distributed.run
    (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT); no process group is created."""
    import torch

    import cugraph_amd as cg

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    single = os.environ.get("CUGRAPH_AMD_MG_TEST_SINGLE_GPU") == "1"  # all ranks share cuda:0 (the IPC path is the same)
    torch.cuda.set_device(0 if single else int(os.environ.get("LOCAL_RANK", str(rank))))
    comm = cg.Comm(f"trav_{os.environ.get('MASTER_PORT', '0')}_{os.getppid() if world > 1 else os.getpid()}", rank, world)
    h = cg.ResourceHandle(comm)
    nv, ne = 1 << args.scale, args.edge_factor << args.scale
    per = (ne + world - 1) // world
    first = min(rank * per, ne)
    count = max(0, min(per, ne - first))
    # global out-degrees (TEPS denominators, root choice): every rank walks the whole generator stream once, in pieces
    outdeg = torch.zeros(nv, dtype=torch.int64, device="cuda")
    piece = 1 << 26
    for f in range(0, ne, piece):
        s_all, _ = cg.generate_rmat_edgelist(h, args.scale, min(piece, ne - f), first_edge=f)
        outdeg += torch.bincount(s_all.to(torch.int64), minlength=nv)
        del s_all
    src, dst = cg.generate_rmat_edgelist(h, args.scale, count, first_edge=first)
    if args.weights == "unit":
        w = torch.ones(count, dtype=torch.float32, device="cuda")
    else:
        w = torch.randint(1, 256, (count,), generator=torch.Generator(device="cuda").manual_seed(1 + rank), device="cuda").to(torch.float32)
    verts = torch.arange(rank, nv, world, dtype=torch.int32, device="cuda")
    t0 = time.perf_counter()
    g = cg.MGGraph(h, cg.GraphProperties(is_multigraph=True), [src], [dst], [w], store_transposed=False, vertices_array=[verts])
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    src_k, dst_k, w_k = src, dst, w  # (kept: the check tests this rank's slice of the edges)
    del src, dst, w, verts
    cand = torch.nonzero(outdeg > 0).flatten().cpu()
    perm = torch.randperm(cand.numel(), generator=torch.Generator().manual_seed(0))[: args.roots]
    roots = cand[perm].tolist()
    outdeg_f = outdeg.to(torch.float64)
    empty = torch.zeros(0, dtype=torch.int32, device="cuda")

    import tempfile

    import numpy as np

    share = Path(tempfile.gettempdir()) / f"cugraph_amd_part_{os.environ.get('MASTER_PORT', '0')}_{os.getppid() if world > 1 else os.getpid()}"
    share.mkdir(parents=True, exist_ok=True)

    def fixed_point_check(kind, v, d, root):
        """OUTSIDE the timed region, on the last root's result: the distances must be THE fixed point of d[v] = min over in-edges (d[u] + w).  Every
        rank saves the (vertex, distance) pairs it got back; the vector is assembled from the files (one node: a shared directory), and every
        rank tests ITS slice of the edge list: no edge may offer less than its destination holds (<=), and over all ranks every reached vertex
        but the root must be ATTAINED by some edge (the ranks' per-vertex minima are merged by rank 0)."""
        np.save(share / f"{kind}_v{rank}.npy", v.cpu().numpy())
        np.save(share / f"{kind}_d{rank}.npy", d.cpu().numpy())
        comm.barrier()
        full = torch.empty(nv, dtype=d.dtype, device="cuda")
        seen = torch.zeros(nv, dtype=torch.int32, device="cuda")
        for r in range(world):
            vv = torch.from_numpy(np.load(share / f"{kind}_v{r}.npy")).cuda().long()
            full[vv] = torch.from_numpy(np.load(share / f"{kind}_d{r}.npy")).cuda()
            seen[vv] += 1
        inf = 2**31 - 1 if kind == "bfs" else float(torch.finfo(torch.float32).max)
        if kind == "bfs":
            full = full.to(torch.int64)
        du = full[src_k.long()]
        cand = torch.where(du == inf, du, du + (1 if kind == "bfs" else w_k))
        too_low = int((cand < full[dst_k.long()]).sum())  # an edge that offers less than its destination holds
        best = torch.full((nv,), inf, dtype=full.dtype, device="cuda")
        best.scatter_reduce_(0, dst_k.long(), cand, "amin", include_self=True)
        np.save(share / f"{kind}_b{rank}.npy", best.cpu().numpy())
        comm.barrier()
        res = None
        if rank == 0:
            for r in range(1, world):
                best = torch.minimum(best, torch.from_numpy(np.load(share / f"{kind}_b{r}.npy")).cuda())
            best[root] = 0
            res = {"not_attained": int((best != full).sum()), "every_vertex_once": bool((seen == 1).all()), "reached": int((full != inf).sum())}
        lows = sum(int(x[0]) for x in comm.allgather_f64([float(too_low)]))
        if rank == 0:
            res.update({"edges_offering_less": lows, "root": int(root), "ok": lows == 0 and res["not_attained"] == 0 and res["every_vertex_once"],
                        "what": "d[v] == min over in-edges (d[u] + w), d[root] == 0 on the last root's result: every rank tests its slice of the edges against the "
                                "assembled distance vector (bit for bit)"})
        comm.barrier()
        return res

    def run(kind):
        times, teps, levels = [], [], []
        last = None
        for i, r in enumerate([roots[0], roots[-1]] + roots):  # two warm-ups (the first call builds the partition and its windows)
            h.sync()
            comm.barrier()
            t0 = time.perf_counter()
            if kind == "bfs":
                d, _, v = cg.bfs(h, g, torch.tensor([r], dtype=torch.int32, device="cuda") if rank == 0 else empty, False, 0, args.predecessors, False)
                unreached = 2**31 - 1
            else:
                v, d, _ = cg.sssp(h, g, int(r), 3.0e38, args.predecessors, False)
                unreached = float(torch.finfo(torch.float32).max)
            h.sync()
            comm.barrier()
            dt = max(x[0] for x in comm.allgather_f64([time.perf_counter() - t0]))
            er = float((outdeg_f[v.long()] * (d != unreached)).sum())
            er = sum(x[0] for x in comm.allgather_f64([er]))
            if i >= 2:
                times.append(dt)
                teps.append(er / dt)
                levels.append(h.last_traversal_stats()["steps"])
            last = (v, d, r)
        hm = len(teps) / sum(1.0 / t for t in teps)
        res = {"ms_mean": round(1e3 * sum(times) / len(times), 3), "ms_median": round(1e3 * sorted(times)[len(times) // 2], 3),
               "ms_all": [round(1e3 * t, 2) for t in times], "ms_min": round(1e3 * min(times), 3), "ms_max": round(1e3 * max(times), 3),
               "mteps_harmonic_mean": round(hm / 1e6, 1), "rounds_mean": round(sum(levels) / len(levels), 1)}
        if not args.no_check:
            chk = fixed_point_check(kind, *last)
            if rank == 0:
                res["check"] = chk
        return res

    out = {"workload": f"partitioned BFS/SSSP through cugraph_graph_create_mg + cugraph_bfs / cugraph_sssp on the library's communicator, RMAT scale {args.scale} "
                       f"edge factor {args.edge_factor}, weights {args.weights}, {world} rank(s)" + (" sharing one GPU" if single and world > 1 else ""),
           "n_gpus": world, "vertices": nv, "edges": ne, "roots": len(roots), "graph_build_s": round(build_s, 3), "scaling": "strong", "transport": "ipc",
           "bfs": run("bfs")}
    if not args.no_sssp:
        out["sssp"] = run("sssp")
    if rank == 0:
        line = json.dumps(out)
        print(line, flush=True)
        if args.out:
            Path(args.out).parent.mkdir(parents=True, exist_ok=True)
            Path(args.out).write_text(line + "\n")
    comm.barrier()
    del g
    h.sync()
    comm.barrier()
    del h
    comm.close()


def main_partitioned(args):
    """Partitioned BFS / SSSP (cugraph_amd/mg_traversal.py): the SAME RMAT graph over all ranks (strong scaling)."""
    import torch
    import torch.distributed as dist

    import cugraph_amd as cg
    from cugraph_amd import mg_traversal as mt

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    single = os.environ.get("CUGRAPH_AMD_MG_TEST_SINGLE_GPU") == "1"  # plumbing check: all ranks share cuda:0, gloo moves the data
    torch.cuda.set_device(0 if single else local_rank)
    if not dist.is_initialized():
        if world == 1 and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
        if single:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    h = cg.ResourceHandle()
    nv, ne = 1 << args.scale, args.edge_factor << args.scale
    per = (ne + world - 1) // world
    first = rank * per
    count = max(0, min(per, ne - first))
    src, dst = cg.generate_rmat_edgelist(h, args.scale, count, first_edge=first)
    if args.weights == "unit":
        w = torch.ones(count, dtype=torch.float32, device="cuda")
    else:
        w = torch.randint(1, 256, (count,), generator=torch.Generator(device="cuda").manual_seed(1 + rank), device="cuda").to(torch.float32)
    if single:
        src, dst, w = src.cpu(), dst.cpu(), w.cpu()
    t0 = time.perf_counter()
    engines = {"bfs": mt.MGTraversal(src, dst, nv, None, "bfs")}
    if not args.no_sssp:
        engines["sssp"] = mt.MGTraversal(src, dst, nv, w, "sssp")
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    part = engines["bfs"].part
    outdeg = torch.bincount(src.to(torch.int64), minlength=nv)
    dist.all_reduce(outdeg)
    cand = torch.nonzero(outdeg > 0).flatten().cpu()
    perm = torch.randperm(cand.numel(), generator=torch.Generator().manual_seed(0))[: args.roots]
    roots = cand[perm].tolist()
    my_outdeg = outdeg[part.local_vertices].to(torch.float64).to(engines["bfs"].engine.device)
    del src, dst

    def run(kind):
        e = engines[kind]
        times, teps, levels = [], [], []
        for i, r in enumerate([roots[0], roots[-1]] + roots):  # two warm-ups (different roots: the collectives size their buffers on first use)
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            _, d, _ = e.run([r], compute_predecessors=args.predecessors)
            torch.cuda.synchronize()
            dist.barrier()
            dt = time.perf_counter() - t0
            unreached = mt.INT32_MAX if kind == "bfs" else mt.FLT_MAX
            er = (my_outdeg * (d.to(my_outdeg.device) != unreached)).sum().reshape(1)
            er = er.cpu() if single else er
            dist.all_reduce(er)
            if i >= 2:
                times.append(dt)
                teps.append(float(er.item()) / dt)
                levels.append(e.levels)
        hm = len(teps) / sum(1.0 / t for t in teps)
        return {"ms_mean": round(1e3 * sum(times) / len(times), 3), "ms_median": round(1e3 * sorted(times)[len(times) // 2], 3),
                "ms_all": [round(1e3 * t, 2) for t in times], "bottom_up_levels_last": getattr(e, "bottom_up_levels", 0),
                "ms_min": round(1e3 * min(times), 3), "ms_max": round(1e3 * max(times), 3),
                "mteps_harmonic_mean": round(hm / 1e6, 1), "rounds_mean": round(sum(levels) / len(levels), 1)}

    out = {"workload": f"partitioned BFS/SSSP, RMAT scale {args.scale} edge factor {args.edge_factor}, weights {args.weights}, {world} rank(s)",
           "n_gpus": world, "vertices": nv, "edges": ne, "roots": len(roots), "graph_build_s": round(build_s, 3), "scaling": "strong",
           "bfs": run("bfs")}
    if not args.no_sssp:
        out["sssp"] = run("sssp")
    if rank == 0:
        print(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


# Which sources a counter figure depends on.  The collector differences two amounts of work, so graph construction cancels: a workload's
# bytes per unit come from its own kernels, the headers they include and the handle / memory pool (core.hip).  Files outside a group
# (graph build, generators, file readers, other algorithms) cannot change that workload's traffic.
TRAFFIC_GROUPS = {
    "pagerank": ["common.hpp", "core.hip", "wave_ops.hpp", "spmv_tiled.hpp", "spmv_tiled.hip", "pagerank.hip"],
    "traversal": ["common.hpp", "core.hip", "traversal_common.hpp", "traversal_bottom_up.hpp", "traversal.hip", "graph.hip", "outer_ids.hip", "prims.hip"],
    "louvain": ["common.hpp", "core.hip", "louvain.hip", "prims.hip"],
}
TRAFFIC_UNMEASURED = ["edgelist.hip", "graph_functions.hip", "mtx.hip", "rmat.hip", "traversal_mg.hip", "traversal_mg_driver.hip", "comm.hpp", "comm.hip",
                      "mg_graph.hpp", "mg_graph.hip"]  # no entry of the counter file runs them per unit of work


def traffic_group_of(key):
    return "pagerank" if key.startswith("pagerank") else "louvain" if key.startswith("louvain") else "traversal"


def kernel_source_hash(group=None):
    """sha256 over the HIP sources of the library (group = None: all of csrc/; else the files of TRAFFIC_GROUPS[group]):
    profiles/traffic_latest.json is only believed when it was measured on this code."""
    import hashlib

    hsh = hashlib.sha256()
    d = ROOT / "cugraph_amd" / "csrc"
    files = sorted(d.glob("*.h*")) if group is None else [d / f for f in sorted(TRAFFIC_GROUPS[group])]
    for f in files:
        hsh.update(f.name.encode())
        hsh.update(f.read_bytes())
    return hsh.hexdigest()[:16]


def counter_traffic(key):
    """(bytes, source) of a workload from profiles/traffic_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/traffic_collect.py;
    FETCH_SIZE doubled per the gfx950 note of the guide).  The file names the source hash it was measured on: a stale file gives
    (None, reason) -- never a number from other kernels."""
    f = ROOT / "profiles" / "traffic_latest.json"
    if not f.exists():
        return None, "profiles/traffic_latest.json absent"
    try:
        t = json.loads(f.read_text())
    except Exception as e:
        return None, f"profiles/traffic_latest.json unreadable: {e!r}"
    grp = traffic_group_of(key)
    have, want = (t.get("group_hashes") or {}).get(grp), kernel_source_hash(grp)
    if have is None:  # a file without per-workload hashes: the hash over all of csrc/ decides
        have, want = t.get("source_hash"), kernel_source_hash()
    if have != want:
        return None, f"STALE: profiles/traffic_latest.json was measured on source hash {have} ({grp}), this build is {want} (re-run tools/traffic_collect.py)"
    e = t.get("entries", {}).get(key)
    if e is None:
        return None, f"profiles/traffic_latest.json has no entry {key}"
    return e.get("hbm_bytes"), f"profiles/traffic_latest.json[{key}] ({t.get('source', 'rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes')}; same source hash; not measured by this run)"


def traversal_bench(cg, h, scale=24, edge_factor=16, n_roots=64, weights="unit", symmetric=False, predecessors=True, do_sssp=True, cpu_scale=20,
                    cpu=True, check=True, both=True):
    """One Graph500-protocol measurement (see the module docstring); returns the dict of the JSON line."""
    import numpy as np
    import torch

    nv, ne = 1 << scale, edge_factor << scale
    src, dst = cg.generate_rmat_edgelist(h, scale, ne)
    if symmetric:
        src, dst = torch.cat([src, dst]), torch.cat([dst, src])
        ne *= 2
    if weights == "unit":
        w = torch.ones(ne, dtype=torch.float32, device="cuda")
    else:
        g_ = torch.Generator(device="cuda").manual_seed(1)
        w = torch.randint(1, 256, (ne,), generator=g_, device="cuda").to(torch.float32)
    verts = torch.arange(nv, dtype=torch.int32, device="cuda")
    t0 = time.perf_counter()
    g = cg.SGGraph(h, cg.GraphProperties(is_multigraph=True, is_symmetric=symmetric), src, dst, w, store_transposed=False, renumber=True,
                   vertices_array=verts)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    if not check:
        del src, dst, w
    # roots: fixed-seed hash order over the vertices with out-edges (the library's own degrees: no framework kernel in the profiles)
    dv, dd = cg.out_degrees(h, g)
    outdeg = torch.zeros(nv, dtype=torch.int64, device="cuda")
    outdeg[dv.to(torch.int64)] = dd.to(torch.int64)
    del dv, dd
    cand = torch.nonzero(outdeg > 0).flatten()
    perm = torch.randperm(cand.numel(), generator=torch.Generator().manual_seed(0))[: n_roots]
    roots = cand[perm.to(cand.device)].to(torch.int32)

    def run(kind, pred=None):
        pred = predecessors if pred is None else pred
        times, edges, steps, inspected, probes, api_times = [], [], [], [], [], []
        from cugraph_amd import pylib
        for i, r in enumerate([roots[0], roots[0]] + list(roots)):  # two warm-ups (the second BFS of a directed graph builds its CSC)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if kind == "bfs":
                d, p, v = cg.bfs(h, g, r.reshape(1).clone(), False, 0, pred, False)
            else:
                v, d, p = cg.sssp(h, g, int(r), 3.0e38, pred, False)
            torch.cuda.synchronize()
            dt_api = time.perf_counter() - t0
            # The timed region is the C entry point (cugraph_bfs / cugraph_sssp: blocking, results complete at return) -- the drop-in boundary, and what
            # the reference's own benchmark brackets (mg_graph500_bfs_test.cu:744-763 times cugraph::bfs, not the copy-out).  The Python mirror around
            # it (bfs.pyx's has_vertex pre-check, three result columns copied into fresh tensors: 3 x 64 MB at RMAT-24) is reported beside it.
            dt = pylib.last_c_call_s["cugraph_bfs" if kind == "bfs" else "cugraph_sssp"]
            st = h.last_traversal_stats()
            if i > 1:
                times.append(dt)
                api_times.append(dt_api)
                edges.append(st["edges_of_reached"] if kind == "bfs" else None)
                steps.append(st["steps"])
                inspected.append(st["edges_inspected"])
                probes.append(st.get("probes", 0))
            last = (v, d)
        run.inspected = inspected
        run.probes = probes
        run.api_ms = round(1e3 * float(np.mean(api_times)), 3)
        return times, edges, steps, last

    HBM_PEAK = 8000.0  # GB/s, /opt/skills/guides/MI355X_MICROARCH.md

    def roofline(kind, e_r, v_r, t_s, pred):
        """algorithmic bytes of one traversal from the reached sub-graph (E_r out-edges of the V_r reached
        vertices): BFS 4 E_r + 8 V_r + 4 V + 4 V_r (+ 4 V + 4 V_r with predecessors) + V / 4; SSSP 8 E_r + 8 V_r + 8 V + 8 V_r."""
        if kind == "bfs":
            b = 4 * e_r + 8 * v_r + 4 * nv + 4 * v_r + (4 * nv + 4 * v_r if pred else 0) + nv // 4
        else:
            b = 8 * e_r + 8 * v_r + 8 * nv + 8 * v_r
        ach = b / t_s / 1e9
        key = f"{kind}_s{scale}_{'sym' if symmetric else weights}{'_pred' if pred else ''}"
        traffic, tsrc = counter_traffic(key)
        return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK, "unit": "GB/s", "frac": round(ach / HBM_PEAK, 4), "traffic": traffic,
                "traffic_source": tsrc, "algorithmic_bytes_per_traversal": int(b), "kernel": "whole traversal (all levels / rounds, harmonic-mean time)"}

    def bellman_check(kind, v, d):
        """OUTSIDE the timed region, on the last root's result: the distances are THE fixed point of d[v] = min over in-edges of
        d[u] + w (w = 1 for BFS; fp32 addition is monotone, so the fixed point is unique and equals Dijkstra's):
        recomputed here with one torch scatter-min over the edge list, compared bit for bit."""
        root = int(roots[-1])
        step = 1 << 27  # edges per pass (bounded temporaries; torch kernels stay below 2^31 elements)
        if kind == "bfs":
            dist = torch.empty(nv, dtype=torch.int64, device="cuda")
            dist[v.to(torch.int64)] = d.to(torch.int64)
            INF = 2147483647
            best = torch.full((nv,), INF, dtype=torch.int64, device="cuda")
            for b0 in range(0, ne, step):
                du = dist[src[b0:b0 + step].long()]
                best.scatter_reduce_(0, dst[b0:b0 + step].long(), torch.where(du == INF, du, du + 1), "amin", include_self=True)
        else:
            dist = torch.empty(nv, dtype=torch.float32, device="cuda")
            dist[v.to(torch.int64)] = d
            INF = torch.finfo(torch.float32).max
            best = torch.full((nv,), INF, dtype=torch.float32, device="cuda")
            for b0 in range(0, ne, step):
                du = dist[src[b0:b0 + step].long()]
                best.scatter_reduce_(0, dst[b0:b0 + step].long(), torch.where(du == INF, du, du + w[b0:b0 + step]), "amin", include_self=True)
        best[root] = 0
        bad = int((best != dist).sum())
        return {"what": "d[v] == min over in-edges (d[u] + w), d[root] == 0, recomputed with torch on the last root's result (bit for bit)",
                "root": root, "violations": bad, "reached": int((dist != INF).sum()), "ok": bad == 0}

    out = {"metric": f"bfs_sssp_mteps_rmat{scale}", "unit": "MTEPS", "n_gpus": 1, "higher_is_better": True, "data": "synthetic",
           "workload": f"RMAT scale {scale} edge factor {edge_factor}{' symmetrised' if symmetric else ''}, {n_roots} roots "
                       f"(fixed-seed choice among the vertices with out-edges; 2 warm-ups), weights {weights}; TEPS = out-edges of the reached "
                       "vertices / time, harmonic mean (mg_graph500_bfs_test.cu:113-114, 757-763)",
           "vertices": nv, "edges": ne, "roots": n_roots, "predecessors": bool(predecessors), "graph_build_s": round(build_s, 3)}
    bt, be, bs, (bv, bd) = run("bfs")
    reached = [int((bd != 2147483647).sum())]  # last root; the reached set of an RMAT giant component hardly varies between roots
    teps = [e / t for e, t in zip(be, bt)]
    hm = len(teps) / sum(1.0 / x for x in teps)
    t_hm = float(np.mean(be)) / hm
    out["bfs"] = {"mean_ms": round(1e3 * float(np.mean(bt)), 3), "min_ms": round(1e3 * float(np.min(bt)), 3), "max_ms": round(1e3 * float(np.max(bt)), 3),
                  "harmonic_mean_mteps": round(hm / 1e6, 1), "mean_levels": float(np.mean(bs)), "mean_edges_of_reached": float(np.mean(be)),
                  "dtype": "int32", "roofline": roofline("bfs", float(np.mean(be)), reached[0], t_hm, predecessors),
                  "timed": "the C entry point cugraph_bfs (blocking; results complete at return)", "python_api_mean_ms": run.api_ms}
    if check:
        out["bfs"]["check"] = bellman_check("bfs", bv, bd)
    out["value"] = out["bfs"]["harmonic_mean_mteps"]

    def other_variant(kind, e_list):
        """the same roots with the predecessor request flipped (the headline carries `predecessors`; the other figure rides beside it)"""
        ot, _, _, _ = run(kind, not predecessors)
        tp = [e / t for e, t in zip(e_list, ot)]
        ohm = len(tp) / sum(1.0 / x for x in tp)
        return {"predecessors": not predecessors, "mean_ms": round(1e3 * float(np.mean(ot)), 3), "harmonic_mean_mteps": round(ohm / 1e6, 1),
                "roofline_frac": roofline(kind, float(np.mean(e_list)), reached[0], float(np.mean(e_list)) / ohm, not predecessors)["frac"]}

    if both:
        out["bfs"]["distance_only" if predecessors else "with_predecessors"] = other_variant("bfs", be)
    if do_sssp:
        st, _, ss, (sv, sd) = run("sssp")
        teps = [e / t for e, t in zip(be, st)]  # same roots: same reached set, scored on the same edge count
        hm = len(teps) / sum(1.0 / x for x in teps)
        t_hm = float(np.mean(be)) / hm
        out["sssp"] = {"mean_ms": round(1e3 * float(np.mean(st)), 3), "min_ms": round(1e3 * float(np.min(st)), 3), "max_ms": round(1e3 * float(np.max(st)), 3),
                       "harmonic_mean_mteps": round(hm / 1e6, 1), "mean_steps": float(np.mean(ss)), "mean_relaxations_per_edge": round(float(np.mean(run.inspected)) / ne, 3),
                       "mean_distance_probes_per_edge": round(float(np.mean(run.probes)) / ne, 3),  # relaxations that went past the L2-resident distance filter to the distance words
                       "dtype": "f32",
                       "roofline": roofline("sssp", float(np.mean(be)), reached[0], t_hm, predecessors),
                       "timed": "the C entry point cugraph_sssp (blocking; results complete at return)", "python_api_mean_ms": run.api_ms}
        if check:
            out["sssp"]["check"] = bellman_check("sssp", sv, sd)
        if both:
            out["sssp"]["distance_only" if predecessors else "with_predecessors"] = other_variant("sssp", be)
        if weights == "unit":  # integer hops: bit-exact against BFS (last root)
            a = torch.empty(nv, dtype=torch.int64, device="cuda"); a[bv.to(torch.int64)] = bd.to(torch.int64)
            b = torch.empty(nv, dtype=torch.float32, device="cuda"); b[sv.to(torch.int64)] = sd
            reach = a != 2147483647
            ok = bool(torch.equal(a[reach].to(torch.float32), b[reach])) and bool((b[~reach] == torch.finfo(torch.float32).max).all())
            out["sssp"]["unit_weight_distances_equal_bfs"] = ok
    if cpu:
        out["cpu_baseline"] = cpu_baseline(min(cpu_scale, scale), weights)
    return out


def main():
    import torch

    import cugraph_amd as cg

    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=24)
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--roots", type=int, default=64, help="Graph500 protocol: 64 roots after the warm-ups")
    ap.add_argument("--cpu-scale", type=int, default=20, help="RMAT scale of the bounded CPU-baseline sample (oracle BFS / Dijkstra)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--out", type=str, default=None, help="also write the JSON line to this file")
    ap.add_argument("--weights", choices=["unit", "int"], default="unit")
    ap.add_argument("--symmetric", action="store_true", help="add the reverse of every edge (Graph500 input)")
    ap.add_argument("--no-sssp", action="store_true")
    ap.add_argument("--predecessors", action="store_true", help="(the default since round 5) the headline figures are WITH predecessors, as python-cugraph asks for "
                                                                  "them and the Graph500 protocol validates them; the distance-only time rides beside")
    ap.add_argument("--single-variant", action="store_true", help="measure only the headline variant (counter collection: one traversal per root and kind)")
    ap.add_argument("--distance-only", action="store_true", help="headline without predecessors (the figures of rounds 1-4); the with-predecessors time rides beside")
    ap.add_argument("--gpus", type=int, default=1, help="> 1 (under torch.distributed.run): the partitioned engine, one rank per GPU")
    ap.add_argument("--partitioned", action="store_true", help="run the partitioned engine even with one rank (comparison with the single-GPU path)")
    ap.add_argument("--transport", choices=["ipc", "rccl"], default=os.environ.get("CUGRAPH_AMD_MG_TRANSPORT", "ipc"),
                    help="partitioned runs: ipc = cugraph_graph_create_mg + cugraph_bfs / cugraph_sssp on the library's communicator (default), rccl = cugraph_amd/mg_traversal.py over torch.distributed")
    args = ap.parse_args()
    if args.gpus > 1 or args.partitioned:
        return main_partitioned_ipc(args) if args.transport == "ipc" else main_partitioned(args)

    torch.cuda.set_device(0)
    h = cg.ResourceHandle()
    out = traversal_bench(cg, h, args.scale, args.edge_factor, args.roots, args.weights, args.symmetric, not args.distance_only, not args.no_sssp,
                          args.cpu_scale, not args.no_cpu_baseline, not args.no_check, not args.single_variant)
    line = json.dumps(out)
    print(line, flush=True)
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(line + "\n")


def cpu_baseline(scale, weights):
    """The oracle (C restatements of bfs_reference / sssp_reference = Dijkstra, single-threaded like the reference's own) on a bounded
    sample: the same generator at a smaller scale, 4 roots."""
    import numpy as np

    from oracle import oracle as orc

    nv, ne = 1 << scale, 16 << scale
    s, d = orc.rmat(scale, ne)
    w = np.ones(ne, np.float32) if weights == "unit" else np.random.default_rng(1).integers(1, 256, ne).astype(np.float32)
    off, idx, ww = orc.coo_to_cs(nv, s, d, w)
    outdeg = np.diff(off)
    roots = np.flatnonzero(outdeg > 0)[:: max(1, int((outdeg > 0).sum()) // 4)][:4]
    res = {}
    for kind in ("bfs", "sssp"):
        teps = []
        for r in roots:
            t0 = time.perf_counter()
            dist = orc.bfs(nv, off, idx, [int(r)])[0] if kind == "bfs" else orc.sssp(nv, off, idx, ww, int(r))[0]
            dt = time.perf_counter() - t0
            reached = dist != (2147483647 if kind == "bfs" else np.finfo(np.float32).max)
            teps.append(float(outdeg[reached].sum()) / dt)
        res[kind] = round(len(teps) / sum(1.0 / t for t in teps) / 1e6, 2)
    return {"value": res["bfs"], "sssp_value": res["sssp"], "unit": "MTEPS", "cores": 1, "kind": "port",
            "sample": f"RMAT-{scale} (same generator, seed 0), {len(roots)} roots, oracle/oracle.c bfs + Dijkstra, harmonic mean"}


if __name__ == "__main__":
    main()
