#!/usr/bin/env python3
"""A/B: gathers in flight per lane (2 vs 4) in the two GAT backward passes, 8 heads x 32 bf16 on the bench graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgll_amd import _lib, ops, synth
dev = torch.device("cuda:0")
g = synth.products_like_graph(dev, seed=0, locality=0.9, self_loops=True, exact=True, permute_ids=True).reorder(seed=0)[0]
heads, fo = 8, 32
h = torch.randn(g.n_cols, heads * fo, device=dev).to(torch.bfloat16).requires_grad_()
s = (0.5 * torch.randn(g.n_rows, heads, device=dev)).requires_grad_()
t = (0.5 * torch.randn(g.n_cols, heads, device=dev)).requires_grad_()
go = torch.randn(g.n_rows, heads * fo, device=dev).to(torch.bfloat16)
def run():
    out = ops.gat_aggregate(g, h, s, t, heads, 0.2, apply_elu=True, mode=0)
    out.backward(go)
for rnd in range(2):
    for u in (2, 4):
        _lib.lib.dgll_hip_debug_tune(7, u)
        run(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): run()
        b.record(); torch.cuda.synchronize()
        print("U=%d fwd+bwd %.2f ms" % (u, a.elapsed_time(b) / 5), flush=True)
