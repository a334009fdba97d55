#!/usr/bin/env python3
"""Weight-gradient products x^T.g at the bench shapes: split-K MFMA kernel (one launch, g read once for a pair) vs the
library formulation it replaces (batched split-K bmm + sum per product)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import dense, ops  # noqa: E402

dev = torch.device("cuda:0")
M = 2449029


def lib_pair(x1, x2, g):
    outs = []
    for x in (x1, x2):
        if x is None:
            continue
        rows = M // 256
        main = rows * 256
        o = torch.bmm(x[:main].view(256, rows, x.shape[1]).transpose(1, 2), g[:main].view(256, rows, g.shape[1])).float().sum(0)
        o += torch.mm(x[main:].t(), g[main:]).float()
        outs.append(o)
    return outs


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


for k1, k2, n in ((256, 256, 256), (100, 100, 256), (256, 0, 47), (256, 0, 256)):
    def feats(cols):
        t = ops.alloc_features(M, cols, torch.bfloat16, dev)
        t.copy_(torch.randn(M, cols, device=dev))
        return t

    x1, g = feats(k1), feats(n)
    x2 = feats(k2) if k2 else None
    t_hip = timed(lambda: dense._grad_weight_hip(x1, x2, g))
    t_lib = timed(lambda: lib_pair(x1, x2, g))
    gb = M * 2 * (k1 + k2 + n) / 1e9
    ref = lib_pair(x1, x2, g)[0]
    got = dense._grad_weight_hip(x1, x2, g)[0]
    err = float((got - ref).abs().max() / ref.abs().max())
    print("K=%d+%d N=%d: MFMA split-K %.3f ms (%.2f TB/s on %.2f GB) | library bmm+sum %.3f ms | max rel diff %.1e" % (
        k1, k2, n, t_hip, gb / t_hip, gb, t_lib, err), flush=True)
    del x1, x2, g
