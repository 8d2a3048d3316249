#!/usr/bin/env python3
"""PageRank throughput on synthetic RMAT (BASELINE.json metric: MTEPS + PageRank iterations/s, % of HBM roofline).

  python bench.py --gpus 1 --steps K --warmup W            (N = 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is ONE power iteration = one pass of the hot path (fused pull-SpMV) over the whole graph, already
resident in HBM.  W untimed steps, then exactly K timed steps bracketed by barrier + synchronize; the maximum over
ranks is reported.  value = E * K / t / 1e6 (MTEPS, whole job).  Workload = RMAT scale 26, edge factor 16,
(a,b,c) = (0.57,0.19,0.19), seed 0, int32 ids, fp32 ranks, alpha 0.85 -- the graph BASELINE.json quotes the metric on;
with N > 1 the SAME graph is partitioned over the ranks ("strong" scaling).
rank 0 prints one JSON line with `roofline` (dominant kernels k_tiled_phase1 + k_tiled_phase2 -- one launch of each per iteration --
timed with HIP events on the library's stream) and `cpu_baseline` (the C oracle with OpenMP on the host cores, bounded sample).
N > 1 (`--transport ipc`, default): cugraph_graph_create_mg + cugraph_pagerank on the library's own communicator (cugraph_amd/mg_capi.py);
`--transport rccl`: the torch.distributed / RCCL orchestration of cugraph_amd/mg.py.
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

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(nv: int, ne: int, weighted: bool = False) -> int:
    """one unweighted fp32/int32 iteration moves 4E + 16V + 4 bytes (+4E weighted)."""
    return 4 * ne + 16 * nv + 4 + (4 * ne if weighted else 0)


def cpu_baseline(scale: int, iters: int):
    """The oracle (C restatement of pagerank_reference, OpenMP) on the host cores, bounded sample."""
    import numpy as np

    from oracle import oracle as orc

    nv, ne = 1 << scale, 16 << scale
    s, d = orc.rmat(scale, ne)
    off, idx, _ = orc.coo_to_cs(nv, d, s)
    t0 = time.perf_counter()
    _, it, _ = orc.pagerank(nv, off, idx, None, 0.85, 0.0, iters, acc64=True)
    dt = time.perf_counter() - t0
    return {"value": round(ne * it / dt / 1e6, 2), "unit": "MTEPS", "cores": orc.num_threads(), "kind": "port",
            "sample": f"RMAT-{scale} (same generator, seed 0), {it} power iterations, oracle/oracle.c with OpenMP; "
                      f"{dt:.2f} s; includes the reference loop's 4 V-length passes per iteration"}


def networkx_baseline(scale: int, iters: int):
    """NetworkX (the CPU library cuGraph's own Python tests compare against, tests/link_analysis/test_pagerank.py) on the
    same generator's edge list, single process.  nx.pagerank raises after exactly max_iter power iterations when tol = 0,
    so the timed region is `iters` iterations (plus its conversion to a scipy matrix, which is part of every call)."""
    try:
        import networkx as nx

        from oracle import oracle as orc
    except Exception as e:  # not installed on this box
        return {"error": repr(e)}
    ne = 16 << scale
    s, d = orc.rmat(scale, ne)
    G = nx.MultiDiGraph()
    G.add_nodes_from(range(1 << scale))
    G.add_edges_from(zip(s.tolist(), d.tolist()))
    t0 = time.perf_counter()
    try:
        nx.pagerank(G, alpha=0.85, max_iter=iters, tol=0.0)
    except nx.PowerIterationFailedConvergence:
        pass
    dt = time.perf_counter() - t0
    return {"value": round(ne * iters / dt / 1e6, 2), "unit": "MTEPS", "cores": 1, "version": nx.__version__,
            "sample": f"RMAT-{scale} (same generator, seed 0), nx.pagerank alpha=0.85 max_iter={iters} tol=0 on a MultiDiGraph; {dt:.2f} s"}


def run_single(args):
    import torch

    import cugraph_amd as cg

    torch.cuda.set_device(0)
    h = cg.ResourceHandle()
    if args.hot_tile is not None:
        h.set_pagerank_hot_tile(args.hot_tile)
    nv, ne = 1 << args.scale, args.edge_factor << args.scale
    t0 = time.perf_counter()
    src, dst = cg.generate_rmat_edgelist(h, args.scale, ne)
    verts = torch.arange(nv, dtype=torch.int32, device="cuda")
    g = cg.SGGraph(h, cg.GraphProperties(is_multigraph=True), src, dst, None, store_transposed=True, renumber=True, vertices_array=verts)
    del src, dst
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    plan = cg.PageRankPlan(h, g, 0.85)
    h.sync()
    plan_s = time.perf_counter() - t0
    # The plan tunes itself (extensions.h cugraph_amd_pagerank_plan_tune): the same plan runs in a +-2.5 % band on this part depending on which physical
    # pages its streamed arrays got, so it times itself on --placements fresh allocations of them and keeps the fastest -- same data, same kernels,
    # same bits.  Part of the plan's construction, outside the timed region like the graph build (tune_s below); --placements 1 turns it off
    # (profiles/r6t_placement_trials.txt, r6u_warmup_or_placement.txt: what it buys, in fresh processes of this command line).
    t0 = time.perf_counter()
    tuned_ms = plan.tune(args.placements) if args.placements > 1 else 0.0
    h.sync()
    run_single.tune = {"placements": args.placements, "seconds": round(time.perf_counter() - t0, 3), "ms_per_iteration_of_the_kept_placement": round(tuned_ms, 4)}
    plan.step(args.warmup)
    h.sync()
    torch.cuda.synchronize()
    # The timed region carries NO per-launch profiling (round 5 had 4 hipEventCreate + 4 hipEventRecord per iteration inside the clock: 12 us of a
    # 123 us step at RMAT-22): per-launch events are off, ONE HIP-event pair on the library's stream brackets the K steps (`region_ms`), the wall
    # clock brackets the same K steps between synchronisations.
    h.kernel_timing(False)
    h.kernel_timing_reset()
    h.kernel_timing_region("timed_region", True)
    t0 = time.perf_counter()
    plan.step(args.steps)  # epsilon = 0: no host synchronisation inside
    h.kernel_timing_region("timed_region", False)
    h.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _, region_ms = h.kernel_timing_get("timed_region")
    # phase breakdown: a SECOND pass of the same K steps, outside the wall clock, with an event pair around every launch
    # (--no-phase-pass: the counter collection of tools/traffic_collect.py differences runs of K and K' steps and wants exactly that many)
    h.kernel_timing(True)
    h.kernel_timing_reset()
    if not args.no_phase_pass:
        plan.step(args.steps)
    h.sync()
    launches, kernel_ms = h.kernel_timing_get("pagerank_spmv") if not args.no_phase_pass else (0, 0.0)
    try:  # the tiled default runs two kernels per iteration: phase 1 (edge stream) + phase 2 (partials -> rows + epilogue)
        launches2, kernel2_ms = h.kernel_timing_get("pagerank_reduce")
    except Exception:
        launches2, kernel2_ms = 0, 0.0
    h.kernel_timing(False)
    check = None if args.no_check else check_result(cg, h, plan, args.scale, ne, nv)
    if check is not None:
        # for the record, outside the timed region: the same iteration when the caller DOES want the L1 change every iteration
        # (what an epsilon > 0 run executes between its stopping tests): previous iterate re-read, pr written every time
        os.environ["CUGRAPH_AMD_PAGERANK_DIFF"] = "1"
        try:
            plan2 = cg.PageRankPlan(h, g, 0.85)
            plan2.step(args.warmup)
            h.sync()
            t0 = time.perf_counter()
            plan2.step(args.steps)
            h.sync()
            check["ms_per_step_tracking_l1_change"] = round((time.perf_counter() - t0) / args.steps * 1e3, 4)
            del plan2
        finally:
            os.environ.pop("CUGRAPH_AMD_PAGERANK_DIFF", None)
        # besides the fixed-count run, the run a user makes -- the ordinary entry point (plan + iterations + result, one
        # call) with epsilon = 1e-6: iterations to converge and end-to-end time of the call (graph construction excluded, as above)
        h.sync()
        t0 = time.perf_counter()
        _, _, conv = cg.pagerank(h, g, None, None, None, None, 0.85, 1e-6, 500, False, fail_on_nonconvergence=False)
        torch.cuda.synchronize()
        api_s = time.perf_counter() - t0
        plan3 = cg.PageRankPlan(h, g, 0.85)
        n_it, conv3 = plan3.step(500, epsilon=1e-6)
        del plan3
        check["converging_run"] = {"epsilon": 1e-6, "iterations": int(n_it), "converged": bool(conv3) and (conv is None or bool(conv)),
                                   "api_call_seconds": round(api_s, 4), "what": "cugraph_pagerank_allow_nonconvergence(alpha 0.85, epsilon 1e-6, max 500): plan "
                                   "construction + iterations (L1 change read back every iteration) + result columns, one call on the built graph"}
    return nv, ne, dt, launches, kernel_ms, build_s, launches2, kernel2_ms, plan_s, check, region_ms


def check_result(cg, h, plan, scale, ne, nv, alpha=0.85):
    """OUTSIDE the timed region: is the vector the timed iterations produced a PageRank iterate?  (a) mass: |sum(pr) - 1|;
    (b) one more iteration of the library must equal ONE explicit fp64 power iteration (torch, on the regenerated edge list)
    of the state the timed region left behind (the update rule of pagerank_impl.cuh:224-327).  The same kernels are compared
    with the CPU oracle over all 20 iterations at RMAT-22 in tests/test_gpu_parity.py; this is the check that the 32-bit
    offsets of the tiled layout still hold at the benchmarked size.  Reference tolerance: pagerank_test.cpp:328-334 (1e-3
    relative); used here: 2e-5 relative on every vertex."""
    import torch

    v, pr_k, _ = plan.result()
    plan.step(1)
    _, pr_k1, _ = plan.result()
    p = torch.empty(nv, dtype=torch.float64, device="cuda")
    p[v.long()] = pr_k.double()
    q = torch.empty(nv, dtype=torch.float64, device="cuda")
    q[v.long()] = pr_k1.double()
    del pr_k, pr_k1, v
    src, dst = cg.generate_rmat_edgelist(h, scale, ne)
    outw = torch.zeros(nv, dtype=torch.float64, device="cuda")
    y = torch.zeros(nv, dtype=torch.float64, device="cuda")
    step = 1 << 27
    for b in range(0, ne, step):
        s = src[b:b + step]
        outw.index_add_(0, s, torch.ones(s.numel(), dtype=torch.float64, device="cuda"))
    xs = p / torch.where(outw == 0, torch.ones_like(outw), outw) * alpha
    for b in range(0, ne, step):
        y.index_add_(0, dst[b:b + step], xs[src[b:b + step]])
    expect = y + (alpha * float(p[outw == 0].sum()) + (1.0 - alpha)) / nv
    rel = float(((q - expect).abs() / expect).max())
    mass = abs(float(p.sum()) - 1.0)
    ok = rel <= 2e-5 and mass <= 1e-4
    return {"mass_err": mass, "one_step_rel": rel, "l1_step": float((q - p).abs().sum()), "tolerance": {"one_step_rel": 2e-5, "mass_err": 1e-4},
            "what": "one more library iteration vs one explicit fp64 power iteration (torch) from the state the timed steps left", "ok": bool(ok)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scale", type=int, default=26)
    ap.add_argument("--placements", type=int, default=8, help="plan.tune(n): the plan keeps the fastest of n placements of its streamed arrays (1 = off)")
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--hot-tile", type=int, default=None, help="x entries staged in LDS per workgroup (default: library choice)")
    ap.add_argument("--cpu-scale", type=int, default=22, help="RMAT scale of the bounded CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the post-timing correctness check of the timed result")
    ap.add_argument("--no-phase-pass", action="store_true", help="skip the second, per-launch-instrumented pass of the K steps (the phase averages are then absent)")
    ap.add_argument("--no-extras", action="store_true", help="skip the bounded BFS / SSSP (RMAT-24) and Louvain (RMAT-22) sub-lines appended at N = 1")
    ap.add_argument("--extra-roots", type=int, default=64, help="roots of the traversal sub-lines (64 = the Graph500 protocol of SURVEY section 8(d))")
    ap.add_argument("--layout", choices=["1d", "2d"], default=os.environ.get("CUGRAPH_AMD_MG_LAYOUT", "1d"),
                    help="N > 1: 1d = destination partition + sparse all-to-all (default), 2d = the reference's R x C layout (all-gather + reduce-scatter)")
    ap.add_argument("--transport", choices=["ipc", "rccl"], default=os.environ.get("CUGRAPH_AMD_MG_TRANSPORT", "ipc"),
                    help="N > 1: ipc = cugraph_graph_create_mg + cugraph_pagerank on the library's communicator (HIP IPC peer writes over xGMI, loop inside the "
                         "library; default), rccl = the torch.distributed orchestration of cugraph_amd/mg.py (RCCL collectives; --layout applies)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, the launch line the driver uses)
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        try:
            import torch

            if torch.cuda.device_count() < args.gpus:  # fewer devices than ranks: refuse, unless the plumbing switch is set
                if env.get("CUGRAPH_AMD_MG_TEST_SINGLE_GPU") != "1":
                    print(json.dumps({"error": f"--gpus {args.gpus} needs {args.gpus} devices, torch sees {torch.cuda.device_count()} "
                                               "(CUGRAPH_AMD_MG_TEST_SINGLE_GPU=1 runs all ranks on cuda:0 as a plumbing check)"}), flush=True)
                    sys.exit(2)
        except ImportError:
            pass
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))
    if args.gpus > 1 or world > 1:
        if args.transport == "ipc":  # both layouts run behind cugraph_graph_create_mg + cugraph_pagerank on the library's communicator (round 6)
            os.environ["CUGRAPH_AMD_MG_LAYOUT"] = args.layout
            from cugraph_amd import mg_capi as mg
        else:
            from cugraph_amd import mg

        try:
            out = mg.bench_main(args)
        except getattr(mg, "IpcUnavailable", ()) as e:
            # every rank raises this together (the ranks agree on the self-check over the bootstrap segment) before any graph exists:
            # the node cannot run the library's communicator; the RCCL orchestration is the other product path of the same kernels
            if rank == 0:
                print(f"bench.py: communicator bring-up failed ({e}); taking --transport rccl", file=sys.stderr, flush=True)
            from cugraph_amd import mg as mg_rccl

            out = mg_rccl.bench_main(args)
            if out is not None:
                out.setdefault("config", {})["transport_note"] = f"--transport ipc failed its bring-up self-check: {e}"[:400]
        if rank == 0 and out is not None:
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(min(args.cpu_scale, args.scale), 10)
            print(json.dumps(out), flush=True)
        try:  # orderly shutdown of the RCCL communicator (every rank has passed bench_main's final barrier)
            import torch.distributed as dist

            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:
            pass
        return

    nv, ne, dt, launches, kernel_ms, build_s, launches2, kernel2_ms, plan_s, check, region_ms = run_single(args)
    value = ne * args.steps / dt / 1e6
    bytes_per_launch = algorithmic_bytes(nv, ne)
    avg1_s = kernel_ms / 1e3 / max(launches, 1)
    avg2_s = kernel2_ms / 1e3 / max(launches2, 1) if launches2 else 0.0
    avg_kernel_s = avg1_s + avg2_s  # one iteration = one launch of each; the algorithmic bytes are those of the iteration
    # `frac` is priced on the wall clock of the timed region (ms_per_step: the conservative figure); `frac_kernels` on the sum of the two
    # HIP-event kernel averages of the second, per-launch-instrumented pass
    step_s = dt / args.steps
    achieved = bytes_per_launch / step_s / 1e9
    achieved_kernels = bytes_per_launch / avg_kernel_s / 1e9 if launches else None
    from bench_traversal import counter_traffic

    traffic, traffic_source = counter_traffic(f"pagerank_s{args.scale}")
    out = {
        "metric": f"pagerank_mteps_rmat{args.scale}", "value": round(value, 1), "unit": "MTEPS", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"PageRank power iteration, RMAT scale {args.scale} edge factor {args.edge_factor} "
                               "(a,b,c)=(0.57,0.19,0.19) seed 0, int32 ids, fp32 ranks, alpha 0.85, CSC with degree-descending renumbering",
                   "vertices": nv, "edges": ne, "parallelism": "1 GPU",
                   "plan": f"tuned before the timed region: the fastest of {args.placements} placements of the plan's streamed arrays (cugraph_amd_pagerank_plan_tune; same data, same kernels, bit-identical results; --placements 1 = untuned)" if args.placements > 1 else "untuned (--placements 1)",
                   "iterations": "fixed count (epsilon = 0, no host synchronisation inside the timed region): the L1 change is not evaluated, so the epilogue does not re-read the previous iterate, and pr -- the result buffer; the iteration state is x = pr/out_w -- is written by the last iteration of the call (DESIGN.md section 3.1, item 5; CUGRAPH_AMD_PAGERANK_DIFF=1 CUGRAPH_AMD_PAGERANK_WRITE_PR=1 restore both)"},
        "iters_per_sec": round(args.steps / dt, 2),
        "graph_build_s": round(build_s, 3), "plan_build_s": round(plan_s, 3), "plan_tune": run_single.tune,
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_basis": "algorithmic bytes of one iteration / ms_per_step (wall clock of the timed region; no per-launch events inside it)",
                     "region_event_ms_per_step": round(region_ms / args.steps, 4),
                     "region_event_basis": "one HIP-event pair on the library's stream around the K timed steps; avg_phase*_ms come from a second pass of K steps outside the wall clock, one event pair per launch",
                     "frac_kernels": None if achieved_kernels is None else round(achieved_kernels / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "traffic_source": traffic_source,
                     "kernel": "k_tiled_phase1 + k_tiled_phase2 (one launch each per iteration)" if launches2 else "k_spmv_flat",
                     "launches": launches, "avg_kernel_ms": round(avg_kernel_s * 1e3, 4),
                     "avg_phase1_ms": round(avg1_s * 1e3, 4), "avg_phase2_ms": round(avg2_s * 1e3, 4),
                     "algorithmic_bytes_per_launch": bytes_per_launch},
    }
    if check is not None:
        out["check"] = check
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(min(args.cpu_scale, args.scale), 10)
        out["cpu_baseline"]["networkx"] = networkx_baseline(min(16, args.scale), 10)
    if not args.no_extras:
        out["extra"] = extras(args)
    print(json.dumps(out), flush=True)


def extras(args):
    """BASELINE.json configs 3 and 5 through the driver, OUTSIDE the PageRank timed region and bounded to a few seconds each:
    BFS + SSSP at RMAT-24 (integer weights 1..255 and unit weights; --extra-roots roots of the Graph500 protocol) and Louvain at
    RMAT-22, each with its own `roofline`, `cpu_baseline` (oracle on a smaller sample) and `check`.  The same functions print the
    stand-alone lines of bench_traversal.py / bench_louvain.py."""
    import torch

    import cugraph_amd as cg
    from bench_louvain import louvain_bench
    from bench_traversal import traversal_bench

    res = {}
    h = cg.ResourceHandle()
    try:
        bounded = "" if args.extra_roots >= 64 else f" [bench.py extra: {args.extra_roots} roots instead of the 64 of SURVEY section 8(d)]"
        # round 5: the headline of every traversal line is WITH predecessors (python-cugraph's default, what the Graph500 protocol validates);
        # `distance_only` inside each line is the same roots without them (the figure rounds 1-4 quoted)
        t = traversal_bench(cg, h, 24, 16, args.extra_roots, "int", False, True, True, 20, not args.no_cpu_baseline, not args.no_check)
        t["workload"] += bounded
        res["bfs"] = dict(t["bfs"], workload=t["workload"], metric="bfs_mteps_rmat24", unit="MTEPS", value=t["bfs"]["harmonic_mean_mteps"],
                          cpu_baseline=None if "cpu_baseline" not in t else {k: v for k, v in t["cpu_baseline"].items() if k != "sssp_value"})
        cb = t.get("cpu_baseline")
        res["sssp"] = dict(t["sssp"], workload=t["workload"], metric="sssp_mteps_rmat24_int_weights", unit="MTEPS", value=t["sssp"]["harmonic_mean_mteps"],
                           cpu_baseline=None if cb is None else dict({k: v for k, v in cb.items() if k not in ("value", "sssp_value")}, value=cb["sssp_value"]))
        del t
        u = traversal_bench(cg, h, 24, 16, args.extra_roots, "unit", False, True, True, 20, False, not args.no_check)
        u["workload"] += bounded
        res["sssp_unit"] = dict(u["sssp"], workload=u["workload"], metric="sssp_mteps_rmat24_unit_weights", unit="MTEPS", value=u["sssp"]["harmonic_mean_mteps"])
        del u
    except Exception as e:  # an extra must never cost the headline line
        res["traversal_error"] = repr(e)
    try:
        res["louvain"] = louvain_bench(cg, h, 22, 8, 2, 0 if args.no_cpu_baseline else 18)
    except Exception as e:
        res["louvain_error"] = repr(e)
    return res


if __name__ == "__main__":
    main()
