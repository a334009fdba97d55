#!/usr/bin/env python3
"""A/B of the row-per-slot SpMM kernel (dgll_hip_debug_tune(5, v): 1 = never, 2 = whenever it applies) on the bench graph
and on a short-row graph shaped like the halo half of an 8-way partition (13 % of the edges kept: ~6.5 per row)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dgll_amd  # noqa: E402
from dgll_amd import _lib, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
g = synth.products_like_graph(dev, seed=0, locality=0.9, exact=True, permute_ids=True).reorder(seed=0)[0]
keep = torch.rand(g.nnz, device=dev) < 0.13
short = dgll_amd.CSRGraph.from_coo(g.row_index()[keep], g.col[keep].long(), None, (g.n_rows, g.n_cols), coalesce=False)
del keep


def timed(graph, x, weighted, reps=6):
    val = torch.rand(graph.nnz, device=dev) if weighted else None
    fn = lambda: ops.spmm_raw(graph, x, val=val, reduce="sum" if weighted else "mean")   # noqa: E731
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


for name, graph in (("bench graph (avg %.0f edges/row)" % (g.nnz / g.n_rows), g), ("short rows (avg %.1f edges/row)" % (short.nnz / short.n_rows), short)):
    for feat in (47, 100, 128, 256):
        x = ops.alloc_features(graph.n_cols, feat, torch.bfloat16, dev, pad_to=64 if feat < 64 else 8)
        x.copy_(torch.randn(graph.n_cols, feat, device=dev))
        for weighted in (False, True):
            res = {}
            for rnd in range(2):
                for mode in (1, 2):
                    _lib.lib.dgll_hip_debug_tune(5, mode)
                    gg = dgll_amd.CSRGraph(graph.rowptr, graph.col, None, graph.n_rows, graph.n_cols, check=False)
                    res.setdefault(mode, []).append(timed(gg, x, weighted))
            print("%-34s F=%-3d %-10s wave-per-row %.3f ms   row-per-slot %.3f ms   (%+.0f %%)" % (
                name, feat, "weighted" if weighted else "unweighted", min(res[1]), min(res[2]), 100.0 * (min(res[2]) / min(res[1]) - 1)), flush=True)
_lib.lib.dgll_hip_debug_tune(5, 0)
