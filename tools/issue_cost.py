#!/usr/bin/env python3
"""Host-side cost of issuing one training step (tiny graph, so the GPU is never the limiter): the per-step floor that
multi-GPU strong scaling runs into once a rank's GPU work shrinks to a few milliseconds."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dgll_amd import dist as ddist, nn as dnn, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
full = synth.products_like_graph(dev, seed=0, n=20000, n_undirected=200000, locality=0.9)
n = full.n_rows
model = dnn.GraphSage(100, [256, 256, 47], None).to(dev)
x = ops.alloc_features(n, 100, torch.bfloat16, dev, pad_to=64); x.copy_(torch.randn(n, 100, device=dev))
labels = torch.randint(0, 47, (n,), device=dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
part = ddist.partition_contiguous(full, 1, 0)
engine = ddist.DistGraph(part, dev)
xl = engine.permute_to_local(x)
racom = ddist.RaCoM(model.parameters(), dev)


def step(kind):
    opt.zero_grad(set_to_none=True)
    out = model.forward_graph(full, x) if kind == "single" else engine.sage_forward(model, xl)
    loss = ops.cross_entropy(out, labels, reduction="sum") / n
    loss.backward()
    if kind != "single":
        racom.all_reduce_and_wait()
    opt.step()


for kind in ("single", "dist-engine"):
    for _ in range(5):
        step(kind)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step(kind)
    issue = (time.perf_counter() - t0) / 50
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / 50
    print("%-12s host issue %.2f ms/step, wall %.2f ms/step (20 k-node graph)" % (kind, issue * 1e3, total * 1e3))
