#!/usr/bin/env python3
"""Interleaved A/B of the SpMM tuning knobs on the headline shape (products-shaped graph, F=256 bf16, mean)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dgll_amd  # noqa: E402
from dgll_amd import _lib, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
feat = int(sys.argv[1]) if len(sys.argv) > 1 else 256
loc = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
base = synth.products_like_graph(dev, seed=0, locality=loc)
x = torch.randn(base.n_cols, feat, device=dev).to(torch.bfloat16)


def tune(**kw):
    keys = {"unroll": 0, "rpw": 1, "flags": 2, "threshold": 3}
    for k, v in kw.items():
        _lib.check(_lib.lib.dgll_hip_debug_tune(keys[k], v), "tune")


def fresh_graph():
    return dgll_amd.CSRGraph(base.rowptr, base.col, None, base.n_rows, base.n_cols, check=False)


def run(g, reps=6):
    ops.spmm_raw(g, x, reduce="mean")
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.spmm_raw(g, x, reduce="mean"); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def degree_sorted(g):
    """Rows permuted by descending degree (timing experiment: longest-processing-time-first dispatch)."""
    from dgll_amd.dist import _csr_rows
    order = torch.argsort(g.degrees(), descending=True, stable=True)
    rp, c, _ = _csr_rows(g.rowptr, g.col, None, order)
    return dgll_amd.CSRGraph(rp, c, None, g.n_rows, g.n_cols, check=False)


configs = [("thr%d rpw%d remap%d" % (t, r, f), dict(unroll=4, rpw=r, flags=f, threshold=t), False)
           for t in (128,) for r in (0, 8) for f in (0, 1)]
sorted_base = degree_sorted(base)
for rnd in range(2):
    for name, kw, srt in configs:
        tune(**kw)
        src = sorted_base if srt else base
        g = dgll_amd.CSRGraph(src.rowptr, src.col, None, src.n_rows, src.n_cols, check=False)
        med, mn = run(g)
        print("round %d  %-30s median %.3f ms  min %.3f ms  (%.2f Gedges/s)" % (rnd, name, med, mn, base.nnz / med / 1e6), flush=True)
