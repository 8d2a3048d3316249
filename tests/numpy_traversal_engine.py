"""CPU engine for the multi-process BFS / SSSP TESTS (test infrastructure only, like oracle/): the contract of
cugraph_amd.mg_traversal.TraversalEngine implemented with numpy, so that the partitioning and the collectives of
MGTraversal can be exercised under gloo without a GPU.  The product package has no CPU engine."""
import numpy as np
import torch

from cugraph_amd.mg_traversal import FLT_MAX, INT32_MAX, TraversalEngine


class NumpyTraversalEngine(TraversalEngine):
    """Reference engine (CPU, numpy): the same contract as the HIP engine, used by the gloo tests."""

    def __init__(self, mode, offsets, indices, weights, n_rows, L, rank, world, row_vertex):
        self.mode, self.n_rows, self.L, self.rank, self.world = mode, n_rows, L, rank, world
        self.off = offsets.numpy().astype(np.int64)
        self.idx = indices.numpy().astype(np.int64)
        self.w = None if weights is None else weights.numpy().astype(np.float32)
        self.row_vertex = row_vertex.numpy().astype(np.int64)
        self.tuple_words = 2 if mode == 0 else 3
        self.device = torch.device("cpu")
        self.in_off = None
        self.sums = (0, 0)

    def enable_bottom_up(self, in_offsets, in_indices, ext_of_g):
        self.in_off = in_offsets.numpy().astype(np.int64)
        self.in_idx = in_indices.numpy().astype(np.int64)
        self.ext_of_g = ext_of_g.numpy().astype(np.int64)
        return True

    def _degree_sums(self, rows):
        if self.in_off is None:
            return 0, 0
        return int((self.off[rows + 1] - self.off[rows]).sum()), int((self.in_off[rows + 1] - self.in_off[rows]).sum())

    def degree_sums(self):
        return self.sums

    def bottom_up(self, front_bits, level):
        front = np.unpackbits(front_bits.numpy().view(np.uint8), bitorder="little").astype(bool)
        mine = self.seen[self.rank * self.L: self.rank * self.L + self.n_rows]
        rows = np.flatnonzero(~mine)
        deg = self.in_off[rows + 1] - self.in_off[rows]
        r = np.repeat(rows, deg)
        pos = np.repeat(self.in_off[rows], deg) + (np.arange(int(deg.sum())) - np.repeat(np.cumsum(deg) - deg, deg))
        g = self.in_idx[pos]
        hit = front[g]
        r, par = r[hit], self.ext_of_g[g[hit]]
        self.new[:] = False
        if r.size:
            order = np.lexsort((par, r))
            r, par = r[order], par[order]
            first = np.ones(r.size, bool)
            first[1:] = r[1:] != r[:-1]
            r, par = r[first], par[first]  # the minimum external id among the frontier parents
            self.dist[r] = level
            self.pred[r] = par
            self.new[r] = True
        self.frontier = np.flatnonzero(self.new[: self.n_rows])
        self.sums = self._degree_sums(self.frontier)
        return int(self.frontier.size)

    def reset(self, source_rows, cutoff, with_pred):
        src = np.unique(source_rows.numpy().astype(np.int64))
        self.with_pred = with_pred
        self.cutoff = np.float32(min(cutoff, FLT_MAX))
        self.frontier = src
        if self.mode == 0:
            self.dist = np.full(self.n_rows, INT32_MAX, np.int64)
            self.pred = np.full(self.n_rows, INT32_MAX, np.int64)
            self.dist[src] = 0
            self.seen = np.zeros(self.L * self.world, bool)
            self.new = np.zeros(self.L, bool)
            self.new[src] = True
        else:
            self.key = np.full(self.n_rows, np.iinfo(np.uint64).max, np.uint64)
            self.key[src] = 0
        return int(src.size)

    def _edges_of_frontier(self):
        f = self.frontier
        deg = self.off[f + 1] - self.off[f]
        u = np.repeat(f, deg)
        pos = np.repeat(self.off[f], deg) + (np.arange(int(deg.sum())) - np.repeat(np.cumsum(deg) - deg, deg))
        return u, pos

    def expand(self):
        u, pos = self._edges_of_frontier()
        g = self.idx[pos]
        if self.mode == 0:
            keep = ~self.seen[g]
            g, par = g[keep], self.row_vertex[u[keep]]
            order = np.lexsort((par, g))
            g, par = g[order], par[order]
            first = np.ones(g.size, bool)
            first[1:] = g[1:] != g[:-1]
            g, par = g[first], par[first]
            out = np.stack([g % self.L, par], axis=1).astype(np.int32)
        else:
            du = (self.key[u] >> np.uint64(32)).astype(np.uint32).view(np.float32)
            nd = (du + self.w[pos]).astype(np.float32)
            keep = nd < self.cutoff
            g, nd, par = g[keep], nd[keep], self.row_vertex[u[keep]] + 1
            key = (nd.view(np.uint32).astype(np.uint64) << np.uint64(32)) | par.astype(np.uint64)
            order = np.lexsort((key, g))
            g, key = g[order], key[order]
            first = np.ones(g.size, bool)
            first[1:] = g[1:] != g[:-1]
            g, key = g[first], key[first]
            out = np.stack([g % self.L, (key >> np.uint64(32)).astype(np.uint32).view(np.int32),
                            (key & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)], axis=1).astype(np.int32)
        counts = np.bincount(g // self.L, minlength=self.world).tolist()  # g is sorted, so the tuples are grouped by owner
        return torch.from_numpy(np.ascontiguousarray(out)), counts

    def apply(self, recv, level):
        r = recv.numpy()
        if self.mode == 0:
            self.new[:] = False
            if r.size:
                rows, par = r[:, 0].astype(np.int64), r[:, 1].astype(np.int64)
                fresh = self.dist[rows] == INT32_MAX
                rows, par = rows[fresh], par[fresh]
                self.dist[rows] = level
                np.minimum.at(self.pred, rows, par)
                self.new[rows] = True
            self.frontier = np.flatnonzero(self.new[: self.n_rows])
            self.sums = self._degree_sums(self.frontier)
        else:
            nxt = np.zeros(self.n_rows, bool)
            if r.size:
                rows = r[:, 0].astype(np.int64)
                key = (r[:, 1].view(np.uint32).astype(np.uint64) << np.uint64(32)) | r[:, 2].view(np.uint32).astype(np.uint64)
                old = self.key.copy()
                np.minimum.at(self.key, rows, key)
                nxt = (self.key >> np.uint64(32)) < (old >> np.uint64(32))
            self.frontier = np.flatnonzero(nxt)
        return int(self.frontier.size)

    def frontier_bits(self):
        return torch.from_numpy(np.packbits(self.new, bitorder="little").view(np.int32).copy())

    def merge_visited(self, gathered):
        self.seen |= np.unpackbits(gathered.numpy().view(np.uint8), bitorder="little").astype(bool)

    def results(self):
        if self.mode == 0:
            pred = np.where(self.pred == INT32_MAX, -1, self.pred).astype(np.int32)
            return torch.from_numpy(self.dist.astype(np.int32)), (torch.from_numpy(pred) if self.with_pred else None)
        reached = self.key != np.iinfo(np.uint64).max
        d = np.where(reached, (self.key >> np.uint64(32)).astype(np.uint32).view(np.float32), np.float32(FLT_MAX)).astype(np.float32)
        p = np.where(reached, (self.key & np.uint64(0xFFFFFFFF)).astype(np.int64) - 1, -1).astype(np.int32)
        return torch.from_numpy(d), (torch.from_numpy(p) if self.with_pred else None)
