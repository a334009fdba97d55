#!/usr/bin/env python3
"""Experiment: does a column-blocked SpMM schedule pay?  The adjacency is split by COLUMN range into S slabs (each slab's
X rows: N/S * F * 2 bytes -- S = 8 at products size and F = 256 is 157 MB, inside the 256 MiB Infinity Cache); slab s
accumulates into Y (increment form).  Prints the sum of the S launches next to the single full launch.
    python tools/slab_probe.py [--locality 0.0] [--permute] [--feat 256]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import ops, synth  # noqa: E402
from dgll_amd.graph import CSRGraph  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--locality", type=float, default=0.0)
ap.add_argument("--permute", action="store_true")
ap.add_argument("--feat", type=int, default=256)
ap.add_argument("--slabs", default="2,4,8,16")
args = ap.parse_args()
dev = torch.device("cuda:0")
g = synth.products_like_graph(dev, seed=0, locality=args.locality, exact=True, permute_ids=args.permute)
n = g.n_rows
x = ops.alloc_features(n, args.feat, torch.bfloat16, dev)
x.copy_(torch.randn(n, args.feat, device=dev).to(torch.bfloat16))
row = g.row_index()
col = g.col.long()
inv = g.inv_degrees()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


g.plan()
full = timed(lambda: ops.spmm_raw(g, x, reduce="mean"))
print("full launch: %.2f ms (%.1f G edges/s)" % (full, g.nnz / full / 1e6))
for S in [int(s) for s in args.slabs.split(",")]:
    parts = []
    for s in range(S):
        lo, hi = n * s // S, n * (s + 1) // S
        m = (col >= lo) & (col < hi)
        sub = CSRGraph.from_coo(row[m], col[m], None, (n, n), coalesce=False)
        sub.plan()
        parts.append(sub)
    y = ops.alloc_features(n, args.feat, torch.bfloat16, dev)

    def run():
        ops.spmm_raw(parts[0], x, reduce="sum", out=y, row_scale=inv)
        for p in parts[1:]:
            ops.spmm_raw(p, x, reduce="sum", out=y, row_scale=inv, accumulate=2)

    t = timed(run)
    ref = ops.spmm_raw(g, x, reduce="mean")
    err = float((y.float() - ref.float()).abs().max())
    print("S = %2d slabs (%.0f MB of X each): %.2f ms total, first slab %.2f ms; max |diff| vs full %.3g" % (
        S, n / S * args.feat * 2 / 1e6, t, timed(lambda: ops.spmm_raw(parts[0], x, reduce="sum", out=y, row_scale=inv)), err))
    del parts
