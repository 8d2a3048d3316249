#!/usr/bin/env python3
"""Golden values for PageRank at the size its headline number is quoted on (BASELINE config 4's graph on one GPU: RMAT scale 26, edge
factor 16, (a, b, c) = (0.57, 0.19, 0.19), seed 0 -- bench.py's input): the C oracle (oracle/oracle.c: orc_pagerank_f32 with fp64
accumulation, pinned to the reference's goldens and to its compiled CPU reference by tests/test_oracle.py) run for a fixed number of
iterations on the whole 1.07 G-edge multigraph.  The vector itself (268 MB) cannot be committed; the fixture keeps its values at 4096
vertices -- the 64 of highest in-degree, 1984 random ones with in-edges, 2048 random ones -- plus the sum and the number of vertices
without in-edges.

    python tests/golden/make_pagerank_fixture.py 26 20      # ~25 GB, a few CPU-minutes per iteration block

Writes tests/golden/pagerank_rmat<scale>.json; test_pagerank_rmat_golden compares the library's result at those vertices."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import oracle as orc  # noqa: E402


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    ef = 16
    nv, ne = 1 << scale, ef << scale
    t0 = time.time()
    s, d = orc.rmat(scale, ne, seed=0)
    indeg = np.bincount(d, minlength=nv)
    off, idx, _ = orc.coo_to_cs(nv, d, s, None)  # CSC: rows are destinations
    del s, d
    print(f"graph: {ne} edges in {time.time() - t0:.0f} s", flush=True)
    t0 = time.time()
    pr, it, _ = orc.pagerank(nv, off, idx, None, 0.85, 0.0, iters, acc64=True)
    cpu_s = time.time() - t0
    assert it == iters
    rng = np.random.default_rng(2026)
    top = np.argsort(-indeg, kind="stable")[:64]
    with_in = np.flatnonzero(indeg > 0)
    pick = np.unique(np.concatenate([top, rng.choice(with_in, 1984, replace=False), rng.choice(nv, 2048, replace=False)])).astype(np.int64)
    out = {"scale": scale, "edge_factor": ef, "seed": 0, "alpha": 0.85, "iterations": iters, "edges": int(ne),
           "vertices_without_in_edges": int((indeg == 0).sum()), "sum": float(pr.astype(np.float64).sum()),
           "vertices": pick.tolist(), "values": [float(x) for x in pr[pick].astype(np.float64)], "in_degrees": indeg[pick].astype(int).tolist(),
           "oracle_cpu_seconds": round(cpu_s, 1), "made_by": "tests/golden/make_pagerank_fixture.py (oracle/oracle.c: orc_pagerank_f32, fp64 accumulation)"}
    f = ROOT / "tests" / "golden" / f"pagerank_rmat{scale}.json"
    f.write_text(json.dumps(out))
    print({k: v for k, v in out.items() if k not in ("vertices", "values", "in_degrees")}, flush=True)


if __name__ == "__main__":
    main()
