"""Multi-GPU through the reference's own entry points: one process per GPU, the library's communicator (csrc/comm.hpp: HIP IPC windows,
peer writes over xGMI) instead of torch.distributed; cugraph_graph_create_mg + the plan form of cugraph_pagerank.  bench.py --gpus N
runs this path (`--transport ipc`, the default); cugraph_amd/mg.py keeps the torch.distributed / RCCL orchestration (`--transport rccl`)."""
from __future__ import annotations

import os
import time

import torch


class IpcUnavailable(RuntimeError):
    """the communicator could not be brought up on this node (every rank raises it together: bench.py then takes --transport rccl)"""


def _bring_up(Comm, ResourceHandle, session, rank, world):
    """Collective: communicator + handle + the library's self-check of every primitive (all-gather, all-to-all-v, all-reduces against
    closed forms; include/cugraph_amd/extensions.h: cugraph_amd_comm_selftest), which also times the two primitives of the iteration
    path.  The ranks agree on the outcome over the host-side bootstrap segment, so a node where HIP IPC between the GPUs does not work
    fails HERE, on every rank, before any graph is built."""
    import ctypes as C

    from . import _capi as capi
    from .pylib import assert_success

    comm = h = None
    link, why = {}, ""
    try:
        comm = Comm(session, rank, world)
        h = ResourceHandle(comm)
        res, err = (C.c_double * 4)(), C.c_void_p()
        assert_success(capi.lib().cugraph_amd_comm_selftest(h.c_resource_handle_ptr, 1 << 20, 8, res, C.byref(err)), err, "cugraph_amd_comm_selftest")
        link = {"device_barrier_us": round(res[0], 2), "peer_push_GBps_per_rank": round(res[1], 1), "ranks_on_distinct_gpus": bool(res[2])}
        ok = 1.0
    except Exception as e:  # noqa: BLE001 -- any failure of the bring-up means "not on this node"
        ok, why = 0.0, f"{type(e).__name__}: {e}"
    try:
        if comm is None:
            raise RuntimeError(why)
        ok = min(x[0] for x in comm.allgather_f64([ok]))
    except Exception as e:  # noqa: BLE001
        ok, why = 0.0, why or f"{type(e).__name__}: {e}"
    if ok != 1.0:
        raise IpcUnavailable(why or "a peer rank failed the communicator self-check")
    return comm, h, link


def _against_single_gpu(args, comm, rank, world, v, x, iterations, nv, ne):
    """OUTSIDE the timed region: the distributed vector after `iterations` iterations against the single-GPU entry point run for the same number
    of iterations on the whole graph by rank 0 (whose result the parity suite pins to the oracle, and bench.py's N = 1 line to an explicit fp64
    step): the ranks' (vertices, values) are assembled through files (one node), every vertex must come back exactly once."""
    import tempfile
    from pathlib import Path

    import numpy as np

    from .pylib import GraphProperties, PageRankPlan, ResourceHandle, SGGraph, generate_rmat_edgelist

    share = Path(tempfile.gettempdir()) / f"cugraph_amd_pr_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'x')}"
    share.mkdir(parents=True, exist_ok=True)
    np.save(share / f"v{rank}.npy", v.cpu().numpy())
    np.save(share / f"x{rank}.npy", x.cpu().numpy())
    comm.barrier()
    res = {"vs_single_gpu": {"ok": True}}
    if rank == 0:
        full = torch.empty(nv, dtype=x.dtype, device="cuda")
        seen = torch.zeros(nv, dtype=torch.int32, device="cuda")
        for r in range(world):
            vv = torch.from_numpy(np.load(share / f"v{r}.npy")).cuda().long()
            full[vv] = torch.from_numpy(np.load(share / f"x{r}.npy")).cuda()
            seen[vv] += 1
        h1 = ResourceHandle()
        src, dst = generate_rmat_edgelist(h1, args.scale, ne)
        g1 = SGGraph(h1, GraphProperties(is_multigraph=True), src, dst, None, store_transposed=True, renumber=True,
                     vertices_array=torch.arange(nv, dtype=torch.int32, device="cuda"))
        del src, dst
        p1 = PageRankPlan(h1, g1, 0.85)
        p1.step(iterations)
        v1, x1, _ = p1.result()
        want = torch.empty(nv, dtype=x1.dtype, device="cuda")
        want[v1.long()] = x1
        diff = (full.double() - want.double()).abs()
        rel = float((diff / want.double().clamp_min(1e-30)).max())
        res["vs_single_gpu"] = {"iterations": int(iterations), "max_abs": float(diff.max()), "max_rel": rel, "every_vertex_once": bool((seen == 1).all()),
                                "ok": bool(float(diff.max()) <= 1e-6 and rel <= 2e-5 and bool((seen == 1).all())),
                                "what": "the assembled distributed vector against cugraph's single-GPU entry point after the same number of iterations on the whole graph"}
        del p1, g1, h1
    comm.barrier()
    return res


def bench_main(args):
    """bench.py --gpus N: strong scaling of the SAME RMAT graph over N ranks.  Launched under torch.distributed.run (RANK / WORLD_SIZE /
    LOCAL_RANK / MASTER_PORT from the environment); no process group is created -- the session name of the communicator comes from
    MASTER_PORT, which the launcher makes unique per job."""
    from .pylib import Comm, GraphProperties, MGGraph, PageRankPlan, ResourceHandle, generate_rmat_edgelist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    single = os.environ.get("CUGRAPH_AMD_MG_TEST_SINGLE_GPU") == "1"  # plumbing check: all ranks share cuda:0 (the IPC path is the same)
    if single:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    session = f"bench_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'x')}"
    comm, h, link = _bring_up(Comm, ResourceHandle, session, rank, world)
    if args.hot_tile is not None:
        h.set_pagerank_hot_tile(args.hot_tile)
    nv, ne = 1 << args.scale, args.edge_factor << args.scale
    per = (ne + world - 1) // world
    first = min(rank * per, ne)
    count = max(0, min(per, ne - first))
    t0 = time.perf_counter()
    src, dst = generate_rmat_edgelist(h, args.scale, count, first_edge=first)
    verts = torch.arange(rank, nv, world, dtype=torch.int32, device="cuda")  # isolated ids are vertices because somebody lists them
    g = MGGraph(h, GraphProperties(is_multigraph=True), [src], [dst], None, store_transposed=True, vertices_array=[verts])
    del src, dst, verts
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    plan = PageRankPlan(h, g, 0.85)  # collective: partition (edge shuffle, exchange plan), tiled structure, windows, first push
    torch.cuda.synchronize()
    plan_s = time.perf_counter() - t0
    plan.step(args.warmup)
    h.sync()
    comm.barrier()
    t0 = time.perf_counter()
    plan.step(args.steps)
    h.sync()
    comm.barrier()
    dt = max(x[0] for x in comm.allgather_f64([time.perf_counter() - t0]))
    # HIP-event timing of this rank's two SpMV kernels over a few extra (untimed) iterations: the per-GPU roofline fraction
    h.kernel_timing(True)
    h.kernel_timing_reset()
    plan.step(3)
    h.sync()
    n1, ms1 = h.kernel_timing_get("pagerank_spmv")
    n2, ms2 = h.kernel_timing_get("pagerank_reduce")
    h.kernel_timing(False)
    # per ITERATION, not per launch: with the exchange in two chunks (pagerank_mgc_plan, more than one rank) each phase is two launches
    p1, p2 = ms1 / 3.0, ms2 / 3.0
    kernel_s = (p1 + p2) / 1e3
    v, x, _ = plan.result()
    mass = float(x.double().sum())
    tot = comm.allgather_f64([mass, float(v.numel()), p1, p2])
    mass_all, rows_all = sum(t[0] for t in tot), int(sum(t[1] for t in tot))
    check = {"mass_err": abs(mass_all - 1.0), "rows": rows_all, "ok": abs(mass_all - 1.0) <= 1e-4 and rows_all == nv,
             "what": "sum of the distributed PageRank vector and number of owned rows over all ranks (the kernels are the single-GPU ones, checked "
                     "against an explicit fp64 step by bench.py at N = 1; tests/test_mg_capi.py checks this path against the oracle)"}
    if not getattr(args, "no_check", False):
        check.update(_against_single_gpu(args, comm, rank, world, v, x, args.warmup + args.steps + 3, nv, ne))
        check["ok"] = bool(check["ok"] and check["vs_single_gpu"]["ok"])
    n_rows = int(v.numel())
    local_edges = int(g.num_local_edges()) if hasattr(g, "num_local_edges") else None
    local_bytes = 4 * (local_edges if local_edges is not None else ne // world) + 16 * n_rows + 4  # this rank's share of 4E + 16V + 4
    out = None
    if rank == 0:
        ms_iter = dt / args.steps * 1e3
        p1m, p2m = max(t[2] for t in tot), max(t[3] for t in tot)
        out = {
            "metric": f"pagerank_mteps_rmat{args.scale}", "value": round(ne * args.steps / dt / 1e6, 1), "unit": "MTEPS", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_iter, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"PageRank power iteration, RMAT scale {args.scale} edge factor {args.edge_factor} (a,b,c)=(0.57,0.19,0.19) seed 0, int32 ids, "
                                   "fp32 ranks, alpha 0.85; cugraph_graph_create_mg + cugraph_pagerank (plan form) on the library's communicator: " +
                                   ("the reference's 2-D R x C layout (rank = c * R + r): x pushed into the column group's windows, partial rows pushed to their owners "
                                    "in the row group and added in slot order, scalars in a [P][4] window, two signals per iteration, iteration loop inside the library"
                                    if os.environ.get("CUGRAPH_AMD_MG_LAYOUT") == "2d" else
                                    "1-D destination partition in global degree order, x pushed into the peers' gather windows over xGMI (HIP IPC), scalars in a [P][4] "
                                    "window, one signal per iteration, iteration loop inside the library"),
                       "vertices": nv, "edges": ne, "parallelism": f"{world} GPUs, 1 process per GPU", "layout": os.environ.get("CUGRAPH_AMD_MG_LAYOUT", "1d"), "transport": "ipc",
                       "backend": "cugraph_amd communicator (HIP IPC peer writes)", "all_ranks_on_one_gpu": single,
                       "link_selftest": link},
            "iters_per_sec": round(args.steps / dt, 2), "graph_build_s": round(build_s, 3), "plan_build_s": round(plan_s, 3),
            "check": check,
            "phase_split_ms": {"phase1": round(p1m, 4), "phase2": round(p2m, 4), "exchange_and_gaps": round(ms_iter - p1m - p2m, 4),
                               "note": "phase 1 / phase 2 = HIP-event averages (max over ranks); the rest of an iteration = push + signal + wait + fold + launch gaps "
                                       "(nothing of it runs on the host: the loop is cugraph_amd_pagerank_plan_step)"},
            "roofline": {"bound": "hbm", "achieved": round(local_bytes / kernel_s / 1e9, 1) if kernel_s > 0 else None, "peak": 8000.0, "unit": "GB/s",
                         "frac": round(local_bytes / kernel_s / 1e9 / 8000.0, 4) if kernel_s > 0 else None, "traffic": None,
                         "kernel": "k_tiled_phase1 + k_tiled_phase2 on rank 0 (per-GPU share of the algorithmic bytes / its kernel time)",
                         "avg_kernel_ms": round(kernel_s * 1e3, 4), "avg_phase1_ms": round(p1, 4), "avg_phase2_ms": round(p2, 4)},
        }
    comm.barrier()
    del plan, g
    h.sync()
    comm.barrier()
    del h
    comm.close()
    return out
