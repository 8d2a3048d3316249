"""SSSP schedules side by side in ONE process on one graph (GPU box): RMAT-24, integer weights 1..255, the bench's roots.
usage: python tools/sssp_sweep.py [scale] [roots] -- prints mean ms, rounds, relaxations per edge per configuration and checks that
every configuration returns the distances of the default one (bit for bit)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cugraph_amd.pylib as cg  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n_roots = int(sys.argv[2]) if len(sys.argv) > 2 else 16
nv, ne = 1 << scale, 16 << scale
h = cg.ResourceHandle()
src, dst = cg.generate_rmat_edgelist(h, scale, ne)
g_ = torch.Generator(device="cuda").manual_seed(1)
w = torch.randint(1, 256, (ne,), generator=g_, device="cuda").to(torch.float32)
verts = torch.arange(nv, dtype=torch.int32, device="cuda")
g = cg.SGGraph(h, cg.GraphProperties(is_multigraph=True), src, dst, w, store_transposed=False, renumber=True, vertices_array=verts)
dv, dd = cg.out_degrees(h, g)
outdeg = torch.zeros(nv, dtype=torch.int64, device="cuda")
outdeg[dv.to(torch.int64)] = dd.to(torch.int64)
cand = torch.nonzero(outdeg > 0).flatten()
perm = torch.randperm(cand.numel(), generator=torch.Generator().manual_seed(0))[:n_roots]
roots = cand[perm.to(cand.device)].to(torch.int32).tolist()
del src, dst, w

KEYS = ("CUGRAPH_AMD_SSSP_MODE", "CUGRAPH_AMD_SSSP_SUBQ", "CUGRAPH_AMD_SSSP_BATCH", "CUGRAPH_AMD_SSSP_DELTA_SCALE", "CUGRAPH_AMD_SSSP_LH", "CUGRAPH_AMD_SSSP_SPLIT_MIN",
        "CUGRAPH_AMD_SSSP_RADIX_DIV")
configs = [{}]
for spec in os.environ.get("SWEEP", "multi:8,dev:8,radix").split(","):
    p = spec.split(":")  # mode[:SUBQ[:BATCH[:DELTA_SCALE]]] or mode:KEY=value:... (CUGRAPH_AMD_SSSP_KEY)
    c = {"CUGRAPH_AMD_SSSP_MODE": p[0]}
    pos = [x for x in p[1:] if "=" not in x]
    for name, v in zip(("SUBQ", "BATCH", "DELTA_SCALE"), pos):
        c["CUGRAPH_AMD_SSSP_" + name] = v
    for x in p[1:]:
        if "=" in x:
            k, v = x.split("=")
            c["CUGRAPH_AMD_SSSP_" + k] = v
    configs.append(c)
configs.append({})  # the default again (drift of the box during the sweep)
ref = None
for c in configs:
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(c)
    times, steps, relax = [], [], []
    ok = True
    for i, r in enumerate([roots[0]] + roots):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        v, d, p = cg.sssp(h, g, int(r), 3.0e38, False, False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        st = h.last_traversal_stats()
        if i > 0:
            times.append(dt)
            steps.append(st["steps"])
            relax.append(st["edges_inspected"] / ne)
        if i == 1:
            if ref is None:
                ref = d.clone()
            else:
                ok = bool(torch.equal(ref, d))
    print(f"{str(c):110s} mean {1e3 * sum(times) / len(times):7.3f} ms  min {1e3 * min(times):6.3f}  rounds {sum(steps) / len(steps):6.1f}  relax/edge {sum(relax) / len(relax):5.2f}  same distances {ok}",
          flush=True)
