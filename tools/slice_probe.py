#!/usr/bin/env python3
"""Would FEATURE-SLICED aggregation pay?  The F=256 SpMM's counter traffic is ~10x compulsory because a community's feature
rows (38 k x 512 B = 19.6 MB) do not fit a 4 MiB L2.  A slice of 32 / 64 columns of the same rows does.  This times the
existing kernel on narrow feature matrices (what one slice of a slice-major layout would be), with and without the
XCD-contiguous row mapping, on the bench graph: slices x time(slice) against time(F=256)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import _lib, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
g = synth.products_like_graph(dev, seed=0, locality=0.9, exact=True, permute_ids=True).reorder(seed=0)[0]
g.plan()


def timed(x, flags, reps=7):
    _lib.check(_lib.lib.dgll_hip_debug_tune(2, flags), "tune")
    ops.spmm_raw(g, x, reduce="mean")
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.spmm_raw(g, x, reduce="mean"); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


base = None
for f in (256, 128, 64, 32):
    x = torch.randn(g.n_cols, f, device=dev).to(torch.bfloat16)
    t0, t1 = timed(x, 0), timed(x, 1)
    if f == 256:
        base = min(t0, t1)
    n = 256 // f
    print("F=%3d: round-robin rows %.3f ms, XCD-contiguous rows %.3f ms  -> %d slices = %.2f / %.2f ms (whole F=256: %.2f)" % (
        f, t0, t1, n, n * t0, n * t1, base), flush=True)
_lib.check(_lib.lib.dgll_hip_debug_tune(2, 0), "tune")
