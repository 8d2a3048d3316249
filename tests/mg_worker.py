"""Worker for the multi-process PageRank tests (launched by tests/test_mg.py, one process per rank).
mode "oracle": gloo on CPU, local compute = CPU engine on the oracle (exercises partitioning + collectives);
mode "hip":    gloo, every rank drives the HIP engine on cuda:0 (exercises the partitioned HIP kernels on one GPU)."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from cugraph_amd import mg  # noqa: E402
from oracle import oracle as orc  # noqa: E402


class OracleEngine(mg.LocalEngine):
    """Same contract as HipLocalEngine, arithmetic by numpy/scipy in fp64 then rounded to fp32 storage."""

    def __init__(self, part, ex, local_dst, weights, outw_local, alpha, initial_local=None):
        import scipy.sparse as sp

        self.part, self.ex, self.alpha = part, ex, float(alpha)
        c = ex.col_of_edge.numpy().astype(np.int64)
        w = np.ones(c.size) if weights is None else weights.numpy().astype(np.float64)
        self.A = sp.csr_matrix((w, (local_dst.numpy().astype(np.int64), c)), shape=(part.n_rows, max(ex.ncols, 1)))
        self.outw = outw_local.numpy().astype(np.float32)
        self.pr = (np.full(part.n_rows, 1.0 / part.nv, np.float32) if initial_local is None else initial_local.numpy().astype(np.float32))
        self.send = torch.zeros(ex.send_elems, dtype=torch.float32)
        self.recv = torch.zeros(ex.recv_elems, dtype=torch.float32)
        self.base = 0.0

    def _pack(self, diff):
        ex, s = self.ex, self.send.numpy()
        div = np.where(self.outw == 0, np.float32(1), self.outw)
        x = (self.pr / div).astype(np.float32)
        idx = ex.send_index.numpy().astype(np.int64)
        dang = float(self.pr[self.outw == 0].astype(np.float64).sum())
        xmax = float(np.abs(x).max()) if x.size else 0.0
        off = first = 0
        for r in range(ex.world):
            n = ex.send_counts[r]
            s[off:off + n] = x[idx[first:first + n]] if x.size else 0
            tail = s[off + n: off + n + ex.tail].view(np.float64)  # (L1 change, dangling mass, max |x|, pad)
            tail[0], tail[1], tail[2] = diff, dang, xmax
            off += n + ex.tail
            first += n

    def start(self):
        self._pack(0.0)

    def reduce_scalars(self, read_back):
        ex, r = self.ex, self.recv.numpy()
        diff = dang = 0.0
        off = 0
        for k in range(ex.world):
            off += ex.recv_counts[k]
            t = r[off: off + ex.tail].copy().view(np.float64)
            diff += t[0]
            dang += t[1]
            off += ex.tail
        dang = np.float32(dang)
        self.base = np.float32((dang * np.float32(self.alpha) + np.float32(1.0 - self.alpha)) / np.float32(self.part.nv))
        return diff, float(dang)

    def local_step(self):
        xc = self.recv.numpy()[self.ex.col_pos.numpy().astype(np.int64)] if self.ex.ncols else np.zeros(1, np.float32)
        x = xc.astype(np.float64) * np.float64(np.float32(self.alpha))
        y = (self.A @ x).astype(np.float32)
        new = (self.base + y).astype(np.float32)
        diff = float(np.abs(new - self.pr).astype(np.float64).sum())
        self.pr = new
        self._pack(diff)

    def values(self):
        return torch.from_numpy(self.pr.copy())


class OracleEngine2D(mg.LocalEngine2D):
    """CPU engine of the 2-D layout (same contract as HipLocalEngine2D): scipy block SpMV in fp64, fp32 storage."""

    def __init__(self, part, local_col, local_row, weights, outw_own, alpha, initial_own=None):
        import scipy.sparse as sp

        self.part, self.alpha = part, float(alpha)
        L, R, Cc = part.L, part.R, part.C
        w = np.ones(local_col.numel()) if weights is None else weights.numpy().astype(np.float64)
        self.A = sp.csr_matrix((w, (local_row.numpy().astype(np.int64), local_col.numpy().astype(np.int64))), shape=(Cc * L, R * L))
        self.outw = np.zeros(L, np.float32)
        self.outw[: part.n_rows] = outw_own.numpy().astype(np.float32)
        self.pr = np.zeros(L, np.float32)
        self.pr[: part.n_rows] = 1.0 / part.nv if initial_own is None else initial_own.numpy().astype(np.float32)
        self.x_own = torch.zeros(L, dtype=torch.float32)
        self.x_cols = torch.zeros(R * L, dtype=torch.float32)
        self.y_part = torch.zeros(Cc * L, dtype=torch.float32)
        self.y_own = torch.zeros(L, dtype=torch.float32)
        self.triple = torch.zeros(4, dtype=torch.float64)
        self.base = np.float32(0)

    def _x_and_scalars(self, diff):
        n = self.part.n_rows
        div = np.where(self.outw == 0, np.float32(1), self.outw)
        x = (self.pr / div).astype(np.float32)
        x[n:] = 0  # padding rows of the last partitions
        self.x_own.numpy()[:] = x
        owned = np.arange(self.part.L) < n
        self.triple[0] = diff
        self.triple[1] = float(self.pr[owned & (self.outw == 0)].astype(np.float64).sum())
        self.triple[2] = float(np.abs(x).max()) if x.size else 0.0

    def start(self):
        self._x_and_scalars(0.0)

    def set_scalars(self, gathered, read_back):
        t = gathered.numpy().reshape(-1, 4)
        diff, dang = float(t[:, 0].sum()), np.float32(t[:, 1].sum())
        self.base = np.float32((dang * np.float32(self.alpha) + np.float32(1.0 - self.alpha)) / np.float32(self.part.nv))
        return diff, float(dang)

    def spmv(self):
        x = self.x_cols.numpy().astype(np.float64) * np.float64(np.float32(self.alpha))
        self.y_part.numpy()[:] = (self.A @ x).astype(np.float32)

    def epilogue(self):
        n = self.part.n_rows
        new = self.pr.copy()
        new[:n] = (self.base + self.y_own.numpy()[:n]).astype(np.float32)
        diff = float(np.abs(new[:n] - self.pr[:n]).astype(np.float64).sum())
        self.pr = new
        self._x_and_scalars(diff)

    def values(self):
        return torch.from_numpy(self.pr[: self.part.n_rows].copy())


def _delay_collectives():
    """Every data collective starts ~10 ms late on the stream it is issued from (a spin kernel in front of it): a consumer that is not
    ordered behind the collective -- the library computing on a stream of its own without a synchronisation -- then reads the previous
    iteration's buffer, every time instead of once in a few runs."""
    def late(fn):
        def call(*a, **k):
            torch.cuda._sleep(20_000_000)
            return fn(*a, **k)
        return call

    for name in ("all_to_all_single", "all_gather_into_tensor", "reduce_scatter_tensor"):
        setattr(dist, name, late(getattr(dist, name)))


def main():
    mode, scale, out_dir = sys.argv[1], int(sys.argv[2]), Path(sys.argv[3])
    eps, max_iter = float(sys.argv[4]), int(sys.argv[5])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    nccl = mode.endswith("_nccl")  # RCCL with device tensors (one rank per GPU; world 1 on the one-GPU box), every collective delayed
    if nccl:
        mode = mode[: -len("_nccl")]
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        _delay_collectives()
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    nv, ne = 1 << scale, 16 << scale
    per = (ne + world - 1) // world
    s, d = orc.rmat(scale, min(per, ne - rank * per), first_edge=rank * per)
    weighted = mode.endswith("w")  # (modes: oracle / hip, optional "2d", optional trailing "w" = weighted, "_rounds")
    if mode.endswith("_rounds"):  # exercise the multi-round path of mg._a2a: every message of more than 64 bytes is cut
        mg._A2A_MAX_BYTES = 64
    w = None
    if weighted:
        w = torch.from_numpy(np.random.default_rng(1).integers(1, 9, size=ne).astype(np.float32)[rank * per: rank * per + s.size].copy())
    two_d = "2d" in mode  # the reference's R x C layout (mg.MGPageRank2D) instead of the 1-D sparse all-to-all
    factory = (OracleEngine2D if two_d else OracleEngine) if mode.startswith("oracle") else None
    if factory is None:
        torch.cuda.set_device(0)
    ts, td = torch.from_numpy(s), torch.from_numpy(d)
    if nccl:
        ts, td, w = ts.cuda(), td.cuda(), (None if w is None else w.cuda())
    if two_d:
        v, x, iters, conv = mg.pagerank_2d(ts, td, nv, weights=w, alpha=0.85, epsilon=eps, max_iterations=max_iter, engine_factory=factory)
    else:
        v, x, iters, conv = mg.pagerank(ts, td, nv, weights=w, alpha=0.85, epsilon=eps, max_iterations=max_iter, engine_factory=factory)
    np.savez(out_dir / f"rank{rank}.npz", v=v.cpu().numpy(), x=x.cpu().numpy(), iters=iters, conv=conv)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
