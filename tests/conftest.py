import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return json.loads((ROOT / "tests" / "golden" / "golden.json").read_text())


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle as o

    o.build()
    return o


def rmat_graph(orc, scale, edge_factor=16, seed=0):
    """Deterministic RMAT edge list (same stream the HIP generator produces)."""
    return orc.rmat(scale, edge_factor << scale, seed=seed)


def int_weights(n, seed=1, lo=1, hi=255):
    rng = np.random.default_rng(seed)
    return rng.integers(lo, hi + 1, size=n).astype(np.float32)
