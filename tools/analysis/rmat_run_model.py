#!/usr/bin/env python3
"""Where the partials of the column-tiled SpMV come from: an analytic model of RMAT.

RMAT with (a, b, c, d) draws every bit of (src, dst) independently: P(src bit = 1) = c + d, P(dst bit = 1) = b + d (0.24 each for
the benchmark's 0.57 / 0.19 / 0.19 / 0.05).  A vertex whose id has k one-bits expects lam_k = E * 0.76^(S-k) * 0.24^k out-edges
and as many in-edges; the degree order of the library is therefore (up to Poisson noise) the order of k, and a source tile of T
consecutive columns is a set of sources with (nearly) one k.  A tile with E_J edges touches sum_k C(S,k) * (1 - exp(-E_J * p_k))
distinct destinations = runs = partial sums (p_k = 0.76^(S-k) * 0.24^k).

Prints, per source class: live sources, tiles, edges per tile, runs per tile, edges per run -- and the totals, which the plan's own
debug line confirms (RMAT-26: 309.9 M runs measured, ~306 M modelled).  Reading: the hottest tile holds 25 % of the EDGES but 5 % of
the RUNS (16 edges per run); two thirds of the runs come from the tiles of classes k = 6..9, where a run has 1.6-4.8 edges; a
"hot corner" kernel that keeps a destination window in LDS would remove long runs that cost next to nothing already."""
import argparse
import math


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=26)
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--tile", type=int, default=32256)
    args = ap.parse_args()
    S, E, T = args.scale, args.edge_factor << args.scale, args.tile
    p1 = 0.24
    pk = [(1 - p1) ** (S - k) * p1 ** k for k in range(S + 1)]
    cnt = [math.comb(S, k) for k in range(S + 1)]
    lam = [E * p for p in pk]

    def distinct(edges):
        return sum(cnt[k] * (1.0 - math.exp(-edges * pk[k])) for k in range(S + 1))

    print(f"RMAT-{S}, E = {E}, tile = {T} columns")
    print(f"{'k':>3} {'live sources':>14} {'tiles':>8} {'edges/tile':>12} {'runs/tile':>12} {'edges/run':>10} {'edges':>14} {'runs':>14}")
    tot_e = tot_r = tot_t = 0.0
    carry_src = carry_edges = 0.0
    for k in range(S + 1):
        live = cnt[k] * (1.0 - math.exp(-lam[k]))
        if live < 1:
            continue
        edges = cnt[k] * lam[k]
        tiles = live / T
        e_tile = edges / max(tiles, 1e-9) if tiles >= 1 else None
        if tiles < 1:  # the hottest classes share tile 0
            carry_src += live
            carry_edges += edges
            continue
        if carry_src:  # fill tile 0 with the head of this class
            take = min(T - carry_src, live)
            e0 = carry_edges + take * lam[k]
            r0 = distinct(e0)
            print(f"{'<' + str(k):>3} {T:>14.0f} {1:>8.0f} {e0:>12.3e} {r0:>12.3e} {e0 / r0:>10.1f} {e0:>14.3e} {r0:>14.3e}   (tile 0)")
            tot_e += e0; tot_r += r0; tot_t += 1
            live -= take; edges -= take * lam[k]; tiles = live / T
            carry_src = 0
        e_tile = edges / tiles
        r_tile = distinct(e_tile)
        print(f"{k:>3} {live:>14.0f} {tiles:>8.1f} {e_tile:>12.3e} {r_tile:>12.3e} {e_tile / r_tile:>10.2f} {edges:>14.3e} {r_tile * tiles:>14.3e}")
        tot_e += edges; tot_r += r_tile * tiles; tot_t += tiles
    print(f"total: {tot_t:.0f} tiles, {tot_e:.4e} edges, {tot_r:.4e} runs ({tot_r / tot_e:.3f} per edge)")


if __name__ == "__main__":
    main()
