#!/usr/bin/env python3
"""Kernel timeline + host time of the GPU side of ONE mini-batch step (config 2 shape: batch 1024, fan-out 25-10-10,
602 features, hidden 256, bf16) on synthetic blocks -- what the consumer of the pipeline does per batch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from dgll_amd import nn as dnn  # noqa: E402
from dgll_amd.graph import CSRGraph  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
fanouts = [25, 10, 10]
sizes = [1024]
for k in fanouts:
    sizes.append(sizes[-1] * k)                     # 1024, 25600, 256000, 2560000 (upper bound: no dedup, as the reference)
feats = [torch.randn(s, 602, device=dev).to(torch.bfloat16) for s in sizes]
labels = torch.randint(0, 41, (1024,), device=dev)
model = dnn.GraphSage(602, [256, 256, 41], fanouts).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)


def step():
    blocks = [CSRGraph.fixed_fanout(sizes[h], fanouts[h], dev) for h in range(3)]      # built per batch, like to_block()
    out = model.forward_sampled(feats, blocks)
    loss = torch.nn.functional.cross_entropy(out.float(), labels)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
host = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 10
print("host issue %.2f ms/step, wall %.2f ms/step" % (host * 1e3, wall * 1e3))
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
print("kernel time %.3f ms, span %.3f ms, %d kernels" % (tot / 1e3, (evs[-1].time_range.end - t0) / 1e3, len(evs)))
from collections import Counter
cnt = Counter(e.name[:70] for e in evs)
for name, c in cnt.most_common(14):
    print("%4d x %s" % (c, name))
cpu = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("aten::")), key=lambda e: -(e.time_range.end - e.time_range.start))
agg = Counter()
for e in cpu:
    agg[e.name] += e.time_range.end - e.time_range.start
print("top host ops (us, inclusive):", [(k, round(v)) for k, v in agg.most_common(12)])
