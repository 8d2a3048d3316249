"""How compressible is dstl16 (the 16-bit tile-local destination of every partial slot that phase 2 reads)?
Slots of a destination tile I are ordered by (source tile J, destination); inside one (I, J) block the destinations ascend,
so a slot can be coded as the gap to the previous slot of its block.  Prints the gap distribution and the bytes per slot of
simple codes, for an RMAT scale (CPU, numpy; same numbering and tiling as the library: descending in-degree, T sources per
tile, TP2 rows per destination tile)."""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle as orc

scale = int(sys.argv[1])
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32256
TP2 = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
nv, ne = 1 << scale, 16 << scale
s, d = orc.rmat(scale, ne)
indeg = np.bincount(d, minlength=nv)
order = np.argsort(-indeg, kind="stable")
pos = np.empty(nv, np.int64)
pos[order] = np.arange(nv)
s, d = pos[s], pos[d]
live = np.zeros(nv, bool)
live[s] = True
col = np.cumsum(live) - 1                       # compact column ids
runs = np.unique((col[s] // T) * nv + d)         # one entry per (source tile, destination)
J, dst = runs // nv, runs % nv
I = dst // TP2                                   # (equal-row tiles here; the library cuts by cost, at most TP2 rows)
o = np.lexsort((dst, J, I))
I, J, dst = I[o], J[o], dst[o]
first = np.ones(dst.size, bool)
first[1:] = (I[1:] != I[:-1]) | (J[1:] != J[:-1])
gap = np.where(first, dst % TP2, dst - np.roll(dst, 1))
P = dst.size
print(f"scale {scale}: P = {P} runs, P/E = {P / ne:.3f}, blocks = {int(first.sum())}, runs per block = {P / first.sum():.1f}")
for b in (4, 6, 8):
    print(f"  gaps < 2^{b}: {np.mean(gap < (1 << b)):.4f}")
esc8 = np.mean(gap >= 255)
print(f"  8-bit gap code with a 16-bit escape: {1 + 2 * esc8:.3f} B/slot (now 2 B/slot); saves {P * (1 - 2 * esc8) / 1e9:.3f} GB per iteration")
nib = np.mean(gap < 15)
print(f"  4-bit gap code with a 16-bit escape: {0.5 + 2 * (1 - nib):.3f} B/slot")
