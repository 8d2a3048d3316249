"""How many (source tile, destination) partial sums does a column-tiled SpMV produce on RMAT?
Sources are renumbered by descending in-degree (what the graph build does), tiles of T consecutive ids."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import oracle as orc

scale = int(sys.argv[1]); Ts = [int(t) for t in sys.argv[2:]] or [16384, 32768]
nv, ne = 1 << scale, 16 << scale
s, d = orc.rmat(scale, ne)
indeg = np.bincount(d, minlength=nv)
order = np.argsort(-indeg, kind="stable")
pos = np.empty(nv, np.int64); pos[order] = np.arange(nv)
s = pos[s]; d = pos[d]
print("scale", scale, "E", ne, "nonempty rows", int((indeg > 0).sum()))
for hot in (16384, 32768, 36864, 65536, 1 << 18, 1 << 20):
    print(f"  fraction of gathers with src < {hot}: {np.mean(s < hot):.3f}")
deg_sorted = indeg[order]
cum = np.cumsum(deg_sorted) / ne
for k in (1, 2, 4, 8, 16, 32, 64):
    print(f"  edges in rows with in-degree <= {k}: {1 - cum[np.searchsorted(-deg_sorted, -k, side='left') - 1] if (deg_sorted > k).any() else 1:.3f}")
for T in Ts:
    key = (s // T) * nv + d
    P = np.unique(key).size
    print(f"T={T}: tiles {nv // T}, partials P={P}  P/E={P / ne:.3f}  bytes/edge two-phase = {2.125 + 10 * P / ne + 16 * nv / ne:.2f} (+tile loads)")
    # hybrid: only the first K tiles are LDS-tiled, the rest direct
