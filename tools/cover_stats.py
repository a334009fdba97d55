#!/usr/bin/env python3
"""How much smaller would a push/pull (bipartite vertex cover) halo exchange be?  For rank r and every peer q: the cut
edges (dst row i on r) <- (src node j on q).  Pull moves one row per distinct j, push one partial row per distinct i;
any vertex cover of the bipartite cut graph is a valid mix.  Greedy degree-peeling cover vs pull-only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import synth  # noqa: E402

dev = torch.device("cuda:0")
loc = float(sys.argv[1]) if len(sys.argv) > 1 else 0.9
g = synth.products_like_graph(dev, seed=0, locality=loc)
n = g.n_rows
row = g.row_index()
col = g.col.long()
for world in (2, 4, 8):
    bounds = [(n * r) // world for r in range(world + 1)]
    tot_pull = tot_cover = 0
    worst_pull = worst_cover = 0
    for r in range(world):
        my = (row >= bounds[r]) & (row < bounds[r + 1])
        pull_r = cover_r = 0
        for q in range(world):
            if q == r:
                continue
            m = my & (col >= bounds[q]) & (col < bounds[q + 1])
            i, j = row[m], col[m]
            pull = int(torch.unique(j).numel())
            # greedy peeling: repeatedly take every vertex (either side) whose remaining degree >= theta
            alive = torch.ones(i.numel(), dtype=torch.bool, device=dev)
            cover = 0
            for theta in (256, 64, 16, 8, 4, 3, 2):
                for side in (i, j):
                    ids, inv, cnt = torch.unique(side[alive], return_inverse=True, return_counts=True)
                    pick = cnt >= theta
                    cover += int(pick.sum())
                    kill = pick[inv]
                    idx = alive.nonzero().flatten()
                    alive[idx[kill]] = False
            cover += int(torch.unique(j[alive]).numel())      # the rest: plain pull
            pull_r += pull
            cover_r += cover
        tot_pull += pull_r
        tot_cover += cover_r
        worst_pull, worst_cover = max(worst_pull, pull_r), max(worst_cover, cover_r)
    print("locality %.1f world %d: rows moved per exchange, pull-only total %d (worst rank %d) | push/pull cover total %d (worst rank %d) -> %.2fx less" % (
        loc, world, tot_pull, worst_pull, tot_cover, worst_cover, tot_pull / max(tot_cover, 1)), flush=True)
