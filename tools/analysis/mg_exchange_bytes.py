"""Per-iteration exchange volume of the 1-D sparse all-to-all of multi-GPU PageRank (cugraph_amd/mg.py) on the real RMAT edge
list, next to the budget of the reference's 2-D scheme.  Single process, CPU, numpy: the partition is
a pure function of the edge list (positions in descending global in-degree order, dealt round-robin: owner = pos % P), a rank
needs the x value of every DISTINCT source among the edges whose destination it owns, and values it owns itself do not
travel.  usage: mg_exchange_bytes.py SCALE [P ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle as orc

scale = int(sys.argv[1])
Ps = [int(p) for p in sys.argv[2:]] or [2, 4, 8]
nv, ne = 1 << scale, 16 << scale
t0 = time.time()
s, d = orc.rmat(scale, ne)
indeg = np.bincount(d, minlength=nv)
order = np.argsort(-indeg, kind="stable")
pos = np.empty(nv, np.int32)
pos[order] = np.arange(nv, dtype=np.int32)
ps, pd = pos[s], pos[d]
del s, d
live = np.zeros(nv, bool)
live[ps] = True
print(f"RMAT-{scale}: V {nv}, E {ne}, sources with out-edges {int(live.sum())} ({live.mean() * 100:.1f} % of V); generated in {time.time() - t0:.0f} s", flush=True)
print("P | recv per rank (values: min / mean / max) | MB per rank per iteration | largest single link MB | dense all-gather MB | 2-D budget MB (R x C)")
for P in Ps:
    owner_dst = (pd % P).astype(np.uint64)
    key = owner_dst * np.uint64(nv) + ps.astype(np.uint64)  # (receiving rank, source position)
    key = np.unique(key)
    r = (key // np.uint64(nv)).astype(np.int64)
    src = (key % np.uint64(nv)).astype(np.int64)
    own = src % P
    remote = own != r
    per_rank = np.bincount(r[remote], minlength=P)
    link = np.bincount(r[remote] * P + own[remote], minlength=P * P).reshape(P, P)  # [receiver, sender]
    R = int(np.floor(np.sqrt(P)))
    while P % R:
        R -= 1
    C = P // R
    budget = ((R - 1) + (C - 1)) * 4 * nv / P / 1e6
    print(f"{P} | {per_rank.min()} / {per_rank.mean():.0f} / {per_rank.max()} | {4 * per_rank.max() / 1e6:.1f} | {4 * link.max() / 1e6:.1f} | "
          f"{(P - 1) / P * 4 * nv / 1e6:.1f} | {budget:.1f} ({R} x {C})", flush=True)
