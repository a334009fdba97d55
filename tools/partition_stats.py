#!/usr/bin/env python3
"""Halo / cut statistics of the bench graph under the contiguous k-way partition (what each rank would exchange)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import dist as ddist  # noqa: E402
from dgll_amd import synth  # noqa: E402

dev = torch.device("cuda:0")
loc = float(sys.argv[1]) if len(sys.argv) > 1 else 0.9
g = synth.products_like_graph(dev, seed=0, locality=loc)
for world in (2, 4, 8):
    for rank in (0, world - 1):
        p = ddist.partition_contiguous(g, world, rank)
        print("locality %.1f world %d rank %d: n_own %d local nnz %d halo nnz %d (%.1f%%) n_halo %d (%.1f MB bf16 x256) n_send %d" % (
            loc, world, rank, p.n_own, p.local.nnz, p.halo.nnz, 100.0 * p.halo.nnz / max(p.nnz, 1), p.n_halo,
            p.n_halo * 512 / 1e6, int(p.send_idx.numel())), flush=True)
