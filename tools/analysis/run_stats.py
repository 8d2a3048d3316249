"""Structure of the column-tiled SpMV on RMAT: per source tile J, edges / runs / run-length classes, and how the runs of a
tile spread over the destination tiles.  CPU only (numpy); same numbering as the library (rows: descending in-degree,
columns: compact ids of the live sources).  usage: run_stats.py SCALE [T] [TP2]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle as orc

scale = int(sys.argv[1])
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32256
TP2 = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
nv, ne = 1 << scale, 16 << scale
t0 = time.time()
s, d = orc.rmat(scale, ne)
indeg = np.bincount(d, minlength=nv)
order = np.argsort(-indeg, kind="stable")
pos = np.empty(nv, np.int32)
pos[order] = np.arange(nv, dtype=np.int32)
s, d = pos[s], pos[d]
live = np.zeros(nv, bool)
live[s] = True
col = (np.cumsum(live) - 1).astype(np.int32)
ncols = int(live.sum())
c = col[s]
del s
key = (c // T).astype(np.uint64) * np.uint64(nv) + d.astype(np.uint64)
del c, d
key.sort()
print(f"scale {scale}: sorted {ne} keys in {time.time() - t0:.0f} s; live columns {ncols}, tiles {(ncols + T - 1) // T}", flush=True)
first = np.ones(ne, bool)
first[1:] = key[1:] != key[:-1]
starts = np.flatnonzero(first)
rl = np.diff(np.append(starts, ne))  # run lengths
rk = key[starts]
J = (rk // np.uint64(nv)).astype(np.int64)
dst = (rk % np.uint64(nv)).astype(np.int64)
P = starts.size
print(f"P = {P}, P/E = {P / ne:.4f}")
nJ = int(J.max()) + 1
EJ = np.bincount(J, weights=rl, minlength=nJ)
PJ = np.bincount(J, minlength=nJ)
classes = [(1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 64), (65, 1 << 30)]
print("run-length classes (all tiles):  class  runs%  edges%")
for lo, hi in classes:
    m = (rl >= lo) & (rl <= hi)
    print(f"   {lo:>3}-{hi if hi < 1 << 30 else 'inf':>4}  {m.sum() / P * 100:6.2f}  {rl[m].sum() / ne * 100:6.2f}")
cumE, cumP = np.cumsum(EJ) / ne, np.cumsum(PJ) / P
print("tile  edges%  cumE%  runs%  cumP%  avg_run  singles%ofruns  maxdst  density_of_prefix")
for j in list(range(0, min(nJ, 12))) + list(range(12, nJ, max(1, nJ // 24))):
    m = J == j
    dj = dst[m]
    md = int(dj.max()) + 1
    sing = float((rl[m] == 1).mean())
    # rows below the 95th percentile destination: how dense is the presence bitmap
    p95 = int(np.quantile(dj, 0.95)) + 1
    dens = float((dj < p95).sum() / p95)
    print(f"{j:4d}  {EJ[j] / ne * 100:6.2f} {cumE[j] * 100:6.2f} {PJ[j] / P * 100:6.2f} {cumP[j] * 100:6.2f}  {EJ[j] / PJ[j]:7.2f}  {sing * 100:6.1f}  {md:9d}  p95row={p95} dens={dens:.3f}")
# blocks: (dest tile of TP2 rows, J)
I = dst // TP2
blk = np.ones(P, bool)
blk[1:] = (J[1:] != J[:-1]) | (I[1:] != I[:-1])
nb = int(blk.sum())
print(f"(I,J) blocks with TP2={TP2}: {nb}, runs per block {P / nb:.2f}")
for tp2 in (8192, 16384, 32768):
    I2 = dst // tp2
    b2 = np.ones(P, bool)
    b2[1:] = (J[1:] != J[:-1]) | (I2[1:] != I2[:-1])
    print(f"   TP2={tp2}: blocks {int(b2.sum())}, runs per block {P / b2.sum():.2f}")
# where do the run DESTINATIONS live: share of runs (slots) by destination row range
for lim in (1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 23, 1 << 24, 1 << 25):
    if lim <= nv:
        print(f"   runs with dst < {lim}: {(dst < lim).mean() * 100:.2f}%   edges: {rl[dst < lim].sum() / ne * 100:.2f}%")
print(f"rows with in-degree > 0: {(indeg > 0).sum()}")
