#!/usr/bin/env python3
"""Golden vector for Louvain at RMAT-22 (the size the single-GPU number of BASELINE config 5's row is quoted on): the C oracle
(oracle/oracle.c: orc_louvain, itself tied to the numpy restatement and to the reference's C-API goldens by tests/test_oracle.py)
on the undirected simple RMAT-22 graph of tests/test_gpu_parity.py: louvain_rmat_input (= bench_louvain.py: undirected_rmat).
Takes ~2.5 minutes of one CPU core, which is why the GPU test compares against this committed fixture instead of re-running it.
Writes tests/golden/louvain_rmat22.json: modularity (exact double), levels, sweeps, number of clusters, sha256 of the int32 cluster column."""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from oracle import oracle as orc  # noqa: E402
from test_gpu_parity import louvain_rmat_input  # noqa: E402

scale = 22
src, dst, w = louvain_rmat_input(orc, scale)
c, q, levels, sweeps = orc.louvain_c(1 << scale, src, dst, w, 100, 1e-7, 1.0)
out = {"scale": scale, "edge_factor": 8, "seed": 5, "directed_edges": int(src.size), "modularity": q, "modularity_hex": float(q).hex(), "levels": levels,
       "sweeps": sweeps, "clusters": int(np.unique(c).size), "clusters_sha256": hashlib.sha256(np.ascontiguousarray(c, np.int32).tobytes()).hexdigest(),
       "made_by": "tests/golden/make_louvain_rmat22.py (oracle/oracle.c: orc_louvain)"}
(Path(__file__).resolve().parent / "louvain_rmat22.json").write_text(json.dumps(out, indent=1) + "\n")
print(out)
