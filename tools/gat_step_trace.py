#!/usr/bin/env python3
"""Kernel timeline of one SpGAT training step (config 4 shape) via torch.profiler."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from dgll_amd import nn as dnn, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
g = synth.products_like_graph(dev, seed=0, locality=0.9, self_loops=True, exact=True, permute_ids=True).reorder(seed=0)[0]
n = g.n_rows
torch.manual_seed(0)
model = dnn.SpGAT(100, 32, 47, dropout=0.0, alpha=0.2, nheads=8).to(dev)
x = ops.alloc_features(n, 100, torch.bfloat16, dev, pad_to=64)
x.copy_(torch.randn(n, 100, device=dev))
labels = torch.randint(0, 47, (n,), device=dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)


def step():
    opt.zero_grad(set_to_none=True)
    out = model(x, g)
    loss = -out.float().gather(1, labels.unsqueeze(1)).sum() / n
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
evs = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA), key=lambda e: e.time_range.start)
t0 = evs[0].time_range.start
tot = 0.0
for e in evs:
    d = e.time_range.end - e.time_range.start
    tot += d
    if d >= 40:
        print("%9.1f us  +%8.1f us  %s" % (e.time_range.start - t0, d, e.name[:100]))
print("kernel time %.3f ms, span %.3f ms" % (tot / 1e3, (evs[-1].time_range.end - t0) / 1e3))
