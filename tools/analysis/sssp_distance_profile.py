"""Where the work of an SSSP on the benchmark graph sits along the distance axis (CPU, scipy; RMAT scale 18 / 20, edge factor 16,
integer weights 1..255 -- the generator and weights of bench_traversal.py).  Prints the distance percentiles of the reached vertices and
of their OUT-EDGES (a vertex weighted by its out-degree): with delta = 32 * mean weight / mean degree = 256 the first near-far window
holds nearly the whole graph, and half of all relaxations start at distances <= 7 -- DESIGN.md section 3.4.
usage: python tools/analysis/sssp_distance_profile.py [scale ...]"""
import sys
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sp
import scipy.sparse.csgraph as cs

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from oracle import oracle as orc  # noqa: E402  (analysis tool: the oracle's generator gives the library's RMAT stream)

for scale in [int(a) for a in sys.argv[1:]] or [18, 20]:
    nv = 1 << scale
    s, d = orc.rmat(scale, 16 << scale, seed=0)
    w = np.random.default_rng(1).integers(1, 256, size=s.size).astype(np.float64)
    order = np.lexsort((w, d, s))  # parallel edges: the lightest one decides
    s2, d2, w2 = s[order], d[order], w[order]
    keep = np.ones(s2.size, bool)
    keep[1:] = (s2[1:] != s2[:-1]) | (d2[1:] != d2[:-1])
    a = sp.csr_matrix((w2[keep], (s2[keep], d2[keep])), shape=(nv, nv))
    outdeg = np.bincount(s, minlength=nv)
    src = int(np.nonzero(outdeg > 0)[0][3])
    t0 = time.time()
    dist = cs.dijkstra(a, indices=src)
    fin = np.isfinite(dist)
    r, deg = dist[fin], outdeg[fin]
    o = np.argsort(r)
    cw = np.cumsum(deg[o]) / deg.sum()
    print(f"RMAT-{scale}: {r.size} vertices reached from {src} ({time.time() - t0:.1f} s)")
    print("  vertices : distance percentiles  1 / 10 / 50 / 90 / 99 / 100 % =", np.percentile(r, [1, 10, 50, 90, 99, 100]).tolist())
    print("  out-edges: distance percentiles 10 / 50 / 90 / 99 %            =", [float(r[o][np.searchsorted(cw, q)]) for q in (0.1, 0.5, 0.9, 0.99)])
