"""Write pattern of phase 1: lane i of a wavefront stores the i-th run total of its 1024 edges to slot(run); slots are
ordered by (destination tile I, source tile J, destination).  For an RMAT scale this prints, per 64-run store instruction,
how many distinct 64-byte sectors / 128-byte lines it touches and the sector efficiency (useful bytes / bytes of the touched
sectors), for several destination-tile heights.  CPU, numpy; numbering and tiling as in the library (descending in-degree,
compact columns, T sources per tile)."""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle as orc

scale = int(sys.argv[1])
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32256
nv, ne = 1 << scale, 16 << scale
s, d = orc.rmat(scale, ne)
indeg = np.bincount(d, minlength=nv)
order = np.argsort(-indeg, kind="stable")
pos = np.empty(nv, np.int64)
pos[order] = np.arange(nv)
s, d = pos[s], pos[d]
live = np.zeros(nv, bool)
live[s] = True
col = np.cumsum(live) - 1
runs = np.unique((col[s] // T) * nv + d)       # phase-1 order: (J, dst)
J, dst = runs // nv, runs % nv
P = runs.size
print(f"scale {scale}: P = {P} runs (P/E = {P / ne:.3f}), source tiles {int(J.max()) + 1}")
for rows in (4096, 8192, 16384, 65536):
    I = dst // rows
    o = np.lexsort((dst, J, I))                # slot order
    slot = np.empty(P, np.int64)
    slot[o] = np.arange(P)
    # store instructions: 64 consecutive runs in phase-1 order (wavefront boundaries ignored: an upper bound on coalescing)
    n = P // 64 * 64
    sl = slot[:n].reshape(-1, 64)
    sec = np.sort(sl * 4 // 64, axis=1)
    nsec = 1 + (np.diff(sec, axis=1) != 0).sum(axis=1)
    lin = np.sort(sl * 4 // 128, axis=1)
    nlin = 1 + (np.diff(lin, axis=1) != 0).sum(axis=1)
    # partially written sectors over the whole buffer do not exist (every slot is written once); what matters is how many
    # sectors ONE instruction spreads over, i.e. how many write requests leave the CU per 256 useful bytes
    print(f"  dst tile {rows:6d} rows: sectors/instr mean {nsec.mean():5.1f} (min 4), lines/instr mean {nlin.mean():5.1f} (min 2), "
          f"useful bytes per touched sector {256 / nsec.mean():5.1f} of 64")
