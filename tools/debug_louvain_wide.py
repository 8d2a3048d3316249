#!/usr/bin/env python3
"""GPU box: Louvain on more than 2^31 directed edges, stage by stage (degrees of the built graph, one level per path, two levels), each result's
modularity recomputed from the edge list in torch.  Usage: python tools/debug_louvain_wide.py [scale] [edge_factor]"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["CUGRAPH_AMD_LOUVAIN_TRACE"] = "1"
import torch  # noqa: E402

import cugraph_amd as cg  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 27
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 9
nv = 1 << scale
h = cg.ResourceHandle()
t0 = time.time()
src, dst = cg.generate_rmat_edgelist(h, scale, ef << scale, seed=5)
s, d = src.to(torch.int64), dst.to(torch.int64)
del src, dst
keep = s != d
key = torch.minimum(s, d)[keep] << 32 | torch.maximum(s, d)[keep]
del s, d, keep
key = torch.unique(key)
lo, hi = (key >> 32).to(torch.int32), (key & 0xFFFFFFFF).to(torch.int32)
del key
wt = (1 + (lo.long() * 7 + hi.long() * 13) % 8).to(torch.float32)
src, dst, w = torch.cat([lo, hi]), torch.cat([hi, lo]), torch.cat([wt, wt])
del lo, hi, wt
ne = int(src.numel())
torch.cuda.synchronize()
print(f"edges {ne} ({ne / 2**31:.3f} x 2^31), built in {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
g = cg.SGGraph(h, cg.GraphProperties(is_symmetric=True), src, dst, w, renumber=False, vertices_array=torch.arange(nv, dtype=torch.int32, device="cuda"))
torch.cuda.synchronize()
print(f"graph created in {time.time() - t0:.1f} s", flush=True)
step = 1 << 28
m = float(w.double().sum())
deg = torch.zeros(nv, dtype=torch.int64, device="cuda")
for a in range(0, ne, step):
    deg += torch.bincount(src[a:a + step].long(), minlength=nv)
try:
    v, od = cg.out_degrees(h, g)
    got = torch.empty(nv, dtype=torch.int64, device="cuda")
    got[v.long()] = od.long()
    print("out-degrees equal torch:", bool(torch.equal(got, deg)), flush=True)
except Exception as e:  # noqa: BLE001
    print("degrees call failed:", e, flush=True)


def q_of(v, c):
    cl = torch.empty(nv, dtype=torch.int64, device="cuda")
    cl[v.to(torch.int64)] = c.to(torch.int64)
    t0 = time.time()
    internal, k = 0, torch.zeros(nv, dtype=torch.int64, device="cuda")
    for b in range(0, ne, step):
        ss, dd, ww = src[b:b + step].long(), dst[b:b + step].long(), w[b:b + step].long()
        internal += int((ww * (cl[ss] == cl[dd])).sum())
        k.index_add_(0, ss, ww)
    torch.cuda.synchronize()
    print(f"  (check: edges pass {time.time() - t0:.1f} s)", flush=True)
    a = torch.zeros(nv, dtype=torch.int64, device="cuda").index_add_(0, cl, k).double()
    torch.cuda.synchronize()
    print(f"  (check: clusters pass {time.time() - t0:.1f} s)", flush=True)
    return internal / m - float((a * a).sum()) / (m * m), int(torch.unique(cl).numel())


def run(tag, levels, **env):
    for k, val in env.items():
        os.environ[k] = val
    torch.cuda.synchronize()
    t0 = time.time()
    v, c, q = cg.louvain(h, g, levels, 1e-7, 1.0, False)
    torch.cuda.synchronize()
    dt = time.time() - t0
    qt, ncl = q_of(v, c)
    print(f"[{tag}] levels<={levels} {dt:.2f} s  Q = {q:.9f}  recomputed {qt:.9f}  diff {abs(q - qt):.2e}  clusters {ncl}  work {h.last_traversal_stats()}", flush=True)
    for k in env:
        del os.environ[k]


which = sys.argv[3].split(",") if len(sys.argv) > 3 else ["default1", "sorted1", "nomid1", "nobig1", "default2"]
if "default1" in which: run("default", 1)
if "sorted1" in which: run("sorted only", 1, CUGRAPH_AMD_LOUVAIN_HASH="0")
if "nomid1" in which: run("chunks + sorted", 1, CUGRAPH_AMD_LOUVAIN_MID="0")
if "nobig1" in which: run("chunks + mid + sorted", 1, CUGRAPH_AMD_LOUVAIN_BIG="0")
if "default2" in which: run("default", 2)
if "full" in which: run("default", 100)
