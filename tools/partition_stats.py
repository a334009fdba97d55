#!/usr/bin/env python3
"""Halo statistics of the bench graph under the engine's partitioner, and what replicating the top-k out-degree nodes on
every rank (the reference's static cache rule, FeatureCache/storage.py:84-98) would take out of the halo.
    python tools/partition_stats.py [--locality 0.9]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import dist as ddist, partition as dpart, reorder as dreorder, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--locality", type=float, default=0.9)
args = ap.parse_args()
dev = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
raw = synth.products_like_graph(dev, seed=0, locality=args.locality, exact=True, permute_ids=True)
n = raw.n_rows
print("graph: %d nodes, nnz %d, %d nodes without edges" % (n, raw.nnz, int((raw.degrees() == 0).sum())))
# the generator's own floor: its planted communities in contiguous ranges (same seed, ids NOT permuted: same edge set up to the
# relabelling), split into equal contiguous parts -- what a perfect community finder + packer would cut
planted = synth.products_like_graph(dev, seed=0, locality=args.locality, exact=True, permute_ids=False)
block = -(-n // 64)
for world in (2, 4, 8):
    pp = (torch.arange(n, device=dev) // block) * world // 64
    q = dpart.partition_quality(planted, pp, world)
    print("planted floor N=%d: cut %.2f %%, edge balance %.3f" % (world, 100 * q["cut"], q["balance"]))
del planted
for world in (2, 4, 8):
    st = {}
    perm, bounds = dpart.partition_and_order(raw, world, seed=0, stats=st)
    print("N=%d partitioner: packed communities cut %.2f %% (balance %.3f) -> refined cut %.2f %% (balance %.3f) in %d passes %s" % (
        world, 100 * st["before"]["cut"], st["before"]["balance"], 100 * st["after"]["cut"], st["after"]["balance"],
        len(st["passes"]), [round(100 * x["cut"], 2) for x in st["passes"]]))
    g = dreorder.relabel(raw, perm)
    deg = g.degrees()
    rank_of_degree = torch.empty(n, dtype=torch.int64, device=dev)
    rank_of_degree[torch.argsort(deg, descending=True, stable=True)] = torch.arange(n, device=dev)
    rows = [bounds[r + 1] - bounds[r] for r in range(world)]
    edges = [int(g.rowptr[bounds[r + 1]] - g.rowptr[bounds[r]]) for r in range(world)]
    print("N=%d: rows per part %s, edges per part (max/mean %.3f)" % (world, rows, max(edges) / (sum(edges) / world)))
    if world == 8:      # every rank's share: the step time of a real run is the slowest rank's
        for r in range(world):
            pr = ddist.partition_contiguous(g, world, r, bounds)
            print("   rank %d: %7d own rows, %8d local + %7d halo edges (%.1f %% of its edges), %6d halo rows (%.2fx own)" % (
                r, pr.n_own, pr.local.nnz, pr.halo.nnz, 100.0 * pr.halo.nnz / max(pr.nnz, 1), pr.n_halo, pr.n_halo / max(pr.n_own, 1)))
            del pr
    p = ddist.partition_contiguous(g, world, 0, bounds)
    e0, e1 = int(g.rowptr[bounds[0]]), int(g.rowptr[bounds[1]])
    col = g.col[e0:e1].long()
    remote = col[(col < bounds[0]) | (col >= bounds[1])]
    halo_ids = torch.unique(remote)
    print("   rank 0: %d own rows, %d local + %d halo edges (cut %.1f %%), %d halo rows (%.2fx own rows)" % (
        p.n_own, p.local.nnz, p.halo.nnz, 100.0 * p.halo.nnz / max(p.nnz, 1), p.n_halo, p.n_halo / p.n_own))
    for k in (0, 16384, 65536, 262144):
        hub_edge = rank_of_degree[remote] < k
        hub_row = rank_of_degree[halo_ids] < k
        print("      top-%-6d replicated: halo rows %7d (-%4.1f %%), halo edges %8d (-%4.1f %%); all-gather %5.1f MB per layer at F=256 bf16" % (
            k, int((~hub_row).sum()), 100.0 * float(hub_row.float().mean()), int((~hub_edge).sum()),
            100.0 * float(hub_edge.float().mean()), k * 512 / 1e6))
    del g, p
