#!/usr/bin/env python3
"""BASELINE config 5 shape on ONE MI355X: RMAT-27 (134 M nodes, > 2^31 nonzeros, F = 128 bf16) -- the int64 row-pointer path
at full size, sized for 288 GB.  Builds the graph in HBM, checks exact properties of the SpMM, times it."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import ops, synth  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 27
dev = torch.device("cuda:0")
t0 = time.time()
g = synth.rmat_graph(scale, 16, seed=0, device=dev, symmetric=False, weighted=False, self_loops=True)
torch.cuda.synchronize()
print("RMAT-%d: n=%d nnz=%d (> 2^31: %s) built in %.1f s, peak %.1f GB" % (scale, g.n_rows, g.nnz, g.nnz > 2**31, time.time() - t0,
                                                                          torch.cuda.max_memory_allocated() / 1e9), flush=True)
torch.cuda.empty_cache()
g.plan()
print("long rows (> 128 nnz): %d" % g.num_long_rows(), flush=True)
F = 128
ones = torch.ones(g.n_cols, F, device=dev, dtype=torch.bfloat16)
y = ops.spmm_raw(g, ones, reduce="mean")
assert bool((y == 1).all()), "mean of ones must be exactly 1 on every row (self loops: no empty rows)"
deg = ops.spmm_raw(g, torch.ones(g.n_cols, 8, device=dev), reduce="sum")[:, 0]
assert torch.equal(deg.long(), g.degrees()) or int((deg.long() - g.degrees()).abs().max()) == 0 or int(g.degrees().max()) > 2**24
print("exactness checks passed (mean(1) == 1 on all %d rows; sum(1) == degree)" % g.n_rows, flush=True)
del ones, y, deg
x = torch.randn(g.n_cols, F, device=dev).to(torch.bfloat16)


def timed(graph, label):
    graph.plan()
    ops.spmm_raw(graph, x, reduce="mean")
    torch.cuda.synchronize()
    reps = 5
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ops.spmm_raw(graph, x, reduce="mean")
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    b_alg = graph.nnz * (F * 2 + 4) + graph.n_rows * (F * 2 + 8)
    print("%-28s SpMM F=%d bf16 mean: %.1f ms, %.2f G edges/s, %.0f GB/s algorithmic (%.0f%% of 8 TB/s), memory in use %.1f GB" % (
        label, F, ms, graph.nnz / ms / 1e6, b_alg / ms / 1e6, b_alg / ms / 1e6 / 80, torch.cuda.memory_allocated() / 1e9), flush=True)


timed(g, "generator's (RMAT) ids:")
# the engine's locality pass: above 2^30 edges label propagation's edge sort does not fit one call, so the pass is the hub-first
# order ("degree", the engine's default there) -- what "lpa" degenerates to on a structure-free graph anyway; "random" for comparison
for method in (["degree", "random"] if g.nnz >= (1 << 30) else ["lpa", "degree", "random"]):
    t0 = time.time()
    g2, perm = g.reorder(method=method)
    torch.cuda.synchronize()
    dt = time.time() - t0
    y2 = ops.spmm_raw(g2, torch.ones(g.n_cols, 8, device=dev), reduce="sum")[:, 0]
    assert torch.equal(y2.long()[:1000], g.degrees()[perm][:1000])
    timed(g2, "reorder(%s), %.1f s:" % (method, dt))
    del g2, perm, y2
    torch.cuda.empty_cache()

# the dense half of config 5 (S = X.W, gcnconv.py:30) at the same row count
from dgll_amd import dense  # noqa: E402

del x
torch.cuda.empty_cache()
xs = torch.randn(g.n_rows, F, device=dev).to(torch.bfloat16)
w = (torch.randn(F, F, device=dev) / F ** 0.5).to(torch.bfloat16)
dense.transform_bf16(xs, w)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(3):
    dense.transform_bf16(xs, w)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 3
print("MFMA transform [%d, %d] . [%d, %d] bf16: %.1f ms, %.2f TB/s" % (g.n_rows, F, F, F, ms, g.n_rows * F * 2 * 2 / ms / 1e9), flush=True)
