#!/usr/bin/env python3
"""Kernel timeline of one training step through the PARTITIONED engine at world size 1 (no halo): what a rank's compute
looks like in the multi-GPU bench, next to tools/step_trace.py (the single-process fused path)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from dgll_amd import dist as ddist, nn as dnn, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
full = synth.products_like_graph(dev, seed=0, locality=0.9)
n = full.n_rows
model = dnn.GraphSage(100, [256, 256, 47], None).to(dev)
x = ops.alloc_features(n, 100, torch.bfloat16, dev, pad_to=64)
x.copy_(torch.randn(n, 100, device=dev))
labels = torch.randint(0, 47, (n,), device=dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
part = ddist.partition_contiguous(full, 1, 0)
engine = ddist.DistGraph(part, dev)
xl = engine.permute_to_local(x)
placed = engine.place_input_halo(xl)
racom = ddist.RaCoM(model.parameters(), dev)


def step():
    opt.zero_grad(set_to_none=True)
    out = engine.sage_forward(model, xl, placed)
    loss = ops.cross_entropy(out, labels, reduction="sum") / n
    loss.backward()
    racom.all_reduce_and_wait()
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
    if d >= 60:
        print("%9.1f us  +%8.1f us  %s" % (e.time_range.start - t0, d, e.name[:100]))
print("kernel time %.3f ms, span %.3f ms" % (tot / 1e3, (evs[-1].time_range.end - t0) / 1e3))
