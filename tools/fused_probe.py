#!/usr/bin/env python3
"""Fused aggregate -> transform kernel next to the SpMM + MFMA transform it replaces (products-sized graph, F = 256 bf16)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import _lib, dense, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
g = synth.products_like_graph(dev, seed=0, locality=0.9, exact=True, permute_ids=True).reorder(seed=0)[0]
g.plan()
n, F = g.n_rows, 256
torch.manual_seed(0)
h = torch.randn(n, F, device=dev).to(torch.bfloat16)
ws = (torch.randn(F, F, device=dev) / 16).to(torch.bfloat16)
wn = (torch.randn(F, F, device=dev) / 16).to(torch.bfloat16)


def timeit(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for u in (4, 8):
    _lib.lib.dgll_hip_debug_tune(0, u)
    print("spmm mean F=256, U=%d: %.3f ms" % (u, timeit(lambda: ops.spmm_raw(g, h, reduce="mean"))), flush=True)
_lib.lib.dgll_hip_debug_tune(0, 4)
agg = ops.spmm_raw(g, h, reduce="mean")
print("transform 256+256 -> 256: %.3f ms" % timeit(lambda: dense.transform_bf16(h, ws.t(), agg, wn.t(), relu=True)), flush=True)
print("fused (agg written):  %.3f ms" % timeit(lambda: dense.sage_fused_forward(g, h, "mean", h, ws.t(), wn.t(), True, keep_agg=True)), flush=True)
print("fused (agg not kept): %.3f ms" % timeit(lambda: dense.sage_fused_forward(g, h, "mean", h, ws.t(), wn.t(), True, keep_agg=False)), flush=True)
print("fused, no self operand (relu(A.X.W), the reference kernel's shape): %.3f ms" % timeit(
    lambda: dense.sage_fused_forward(g, h, "mean", None, None, wn.t(), True, keep_agg=False)), flush=True)
