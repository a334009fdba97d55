#!/usr/bin/env python3
"""Config 4 shape: 2-layer GAT (8 heads x 32) on the products-shaped graph -- fused edge-softmax + aggregation kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
loc = float(sys.argv[1]) if len(sys.argv) > 1 else 0.9
g = synth.products_like_graph(dev, seed=0, locality=loc, self_loops=True)
heads, fo = 8, 32
n = g.n_rows


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for dtype in (torch.bfloat16, torch.float32):
    esz = 2 if dtype == torch.bfloat16 else 4
    h = torch.randn(n, heads * fo, device=dev).to(dtype).requires_grad_()
    s = torch.randn(n, heads, device=dev, requires_grad=True)
    tt = torch.randn(n, heads, device=dev, requires_grad=True)
    go = torch.randn(n, heads * fo, device=dev).to(dtype)
    g.transpose()
    fwd = t(lambda: ops.gat_aggregate(g, h, s, tt, heads, 0.2, apply_elu=True, mode=0))
    out = ops.gat_aggregate(g, h, s, tt, heads, 0.2, apply_elu=True, mode=0)

    def bwd():
        torch.autograd.grad(out, (h, s, tt), go, retain_graph=True)

    b = t(bwd)
    F = heads * fo
    b_alg = g.nnz * (F * esz + 4 + 4 * heads) + n * (F * esz + 8 + 8 * heads)
    print("%s GAT 8x32 locality %.1f nnz %d: fwd %.3f ms (%.2f Gedges/s, %.0f GB/s alg = %.0f%% of 8 TB/s) | bwd (2 gather passes) %.3f ms" % (
        str(dtype).replace("torch.", ""), loc, g.nnz, fwd, g.nnz / fwd / 1e6, b_alg / fwd / 1e6, b_alg / fwd / 1e6 / 80, b), flush=True)
