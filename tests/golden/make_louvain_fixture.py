#!/usr/bin/env python3
"""Golden vectors for Louvain at the sizes its timings are quoted on: the C oracle (oracle/oracle.c: orc_louvain, itself tied to the
numpy restatement and to the reference's C-API goldens by tests/test_oracle.py) on the undirected simple RMAT graph of
tests/test_gpu_parity.py: louvain_rmat_input (= bench_louvain.py: undirected_rmat), built here with one in-place sort of packed
(src, dst) keys instead of unique + lexsort so that RMAT-26 (1.06 G directed edges) fits a 62 GB host.

    python tests/golden/make_louvain_fixture.py 24      # ~15 CPU-minutes
    python tests/golden/make_louvain_fixture.py 26      # hours of one core, ~45 GB

Writes tests/golden/louvain_rmat<scale>.json: modularity (exact double), levels, sweeps, number of clusters, sha256 of the int32
cluster column.  The GPU tests compare against these committed files (test_louvain_rmat_golden)."""
import hashlib
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import oracle as orc  # noqa: E402


def undirected_rmat_lean(scale, edge_factor=8, seed=5):
    """the graph of louvain_rmat_input, sorted by (src, dst): distinct {lo, hi} pairs without self-loops, both directions, w = 1 + (7 lo + 13 hi) mod 8"""
    s, d = orc.rmat(scale, edge_factor << scale, seed=seed)
    keep = s != d
    s, d = s[keep], d[keep]
    lo = np.minimum(s, d).astype(np.int64)
    hi = np.maximum(s, d).astype(np.int64)
    del s, d, keep
    lo <<= 32
    lo |= hi
    del hi
    key = np.unique(lo)  # distinct undirected pairs, ascending by (lo, hi)
    del lo
    rev = (key & 0xFFFFFFFF) << 32 | (key >> 32)
    both = np.concatenate([key, rev])
    del key, rev
    both.sort()  # (src, dst) order; no duplicates (lo < hi strictly, so a pair and its reverse differ)
    src = (both >> 32).astype(np.int32)
    dst = (both & 0xFFFFFFFF).astype(np.int32)
    del both
    a, b = np.minimum(src, dst).astype(np.int64), np.maximum(src, dst).astype(np.int64)
    a *= 7
    b *= 13
    a += b
    del b
    a %= 8
    a += 1
    w = a.astype(np.float32)
    return src, dst, w


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    t0 = time.time()
    src, dst, w = undirected_rmat_lean(scale)
    t1 = time.time()
    print(f"graph: {src.size} directed edges in {t1 - t0:.0f} s", flush=True)
    c, q, levels, sweeps = orc.louvain_c(1 << scale, src, dst, w, 100, 1e-7, 1.0)
    t2 = time.time()
    out = {"scale": scale, "edge_factor": 8, "seed": 5, "directed_edges": int(src.size), "modularity": q, "modularity_hex": float(q).hex(), "levels": levels,
           "sweeps": sweeps, "clusters": int(np.unique(c).size), "clusters_sha256": hashlib.sha256(np.ascontiguousarray(c, np.int32).tobytes()).hexdigest(),
           "oracle_cpu_seconds": round(t2 - t1, 1), "made_by": "tests/golden/make_louvain_fixture.py (oracle/oracle.c: orc_louvain)"}
    (Path(__file__).resolve().parent / f"louvain_rmat{scale}.json").write_text(json.dumps(out, indent=1) + "\n")
    print(out)


if __name__ == "__main__":
    main()
