#!/usr/bin/env python3
"""Kernel timeline of rank 0's step at world size N (stand-in transport, see tools/scaling_model.py).
usage: scaling_trace.py [N] [sage|gat]      (gat: config 4's partitioned SpGAT step; N = 1 with gat: the single-GPU SpGAT step)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from dgll_amd import dist as ddist, nn as dnn, ops, partition as dpart, reorder as dreorder, synth  # noqa: E402
from dgll_amd.optim import FlatAdam  # noqa: E402
from scaling_model import NullExchange  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
workload = sys.argv[2] if len(sys.argv) > 2 else "sage"
dev = torch.device("cuda:0")
raw = synth.products_like_graph(dev, seed=0, locality=0.9, exact=True, permute_ids=True, self_loops=workload == "gat")
n = raw.n_rows
if world > 1:
    perm, bounds = dpart.partition_and_order(raw, world, seed=0)
    full = dreorder.relabel(raw, perm)
else:
    full, perm = raw.reorder(seed=0)
    bounds = [0, n]
del raw
model = (dnn.SpGAT(100, 32, 47, dropout=0.0, alpha=0.2, nheads=8) if workload == "gat" else dnn.GraphSage(100, [256, 256, 47], None)).to(dev)
params = list(model.parameters())
opt = FlatAdam(params, lr=1e-3)
racom = ddist.RaCoM(params, dev, flat=opt)
rank = int(os.environ.get("MODEL_RANK", "0")) % world
part = ddist.partition_contiguous(full, world, rank, bounds)
engine = ddist.DistGraph(part, dev)
engine.exchange = NullExchange(part)
x = ops.alloc_features(part.n_own, 100, torch.bfloat16, dev, pad_to=64)
x.copy_(torch.randn(part.n_own, 100, device=dev))
labels = torch.randint(0, 47, (part.n_own,), device=dev)
placed = engine.place_input_halo(x)


if world == 1:
    full.plan(); full.transpose()[0].plan()


def step():
    opt.zero_grad(set_to_none=True)
    if workload == "gat":
        act = model.forward_activations(x, full) if world == 1 else engine.spgat_forward(model, x, placed, activations=True)
        loss = ops.cross_entropy(act, labels, reduction="sum") * (world / n)
    else:
        out = model.forward_graph(full, x) if world == 1 else engine.sage_forward(model, x, placed)
        loss = ops.cross_entropy(out, labels, reduction="sum", fold_relu=True) * (world / n)
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
    if d >= 40:
        print("%9.1f us  +%8.1f us  %s" % (e.time_range.start - t0, d, e.name[:100]))
print("N=%d rank %d kernel time %.3f ms, span %.3f ms, %d kernels" % (world, rank, tot / 1e3, (evs[-1].time_range.end - t0) / 1e3, len(evs)))
small = {}
for e in evs:
    d = e.time_range.end - e.time_range.start
    if d < 40:
        k = small.setdefault(e.name[:230], [0, 0.0])
        k[0] += 1; k[1] += d
print("kernels under 40 us: %d, %.3f ms in total" % (sum(v[0] for v in small.values()), sum(v[1] for v in small.values()) / 1e3))
for name, (cnt, us) in sorted(small.items(), key=lambda kv: -kv[1][1])[:24]:
    print("  %3d x %7.1f us  %s" % (cnt, us, name))
