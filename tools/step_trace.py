#!/usr/bin/env python3
"""Kernel-by-kernel timeline of ONE bench training step (launch order, duration) via torch.profiler -- to see which
elementwise passes surround the SpMM / GEMM kernels.  Usage: python tools/step_trace.py [--locality 0.9]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dgll_amd import nn as dnn, ops, synth  # noqa: E402
from dgll_amd.optim import FlatAdam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--locality", type=float, default=0.9)
    ap.add_argument("--nodes", type=int, default=2449029)
    ap.add_argument("--edges", type=int, default=61859140)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    g = synth.products_like_graph(dev, seed=0, n=args.nodes, n_undirected=args.edges, locality=args.locality, exact=True,
                                  permute_ids=True).reorder(seed=0)[0]          # bench.py's default workload
    n = g.n_rows
    model = dnn.GraphSage(100, [256, 256, 47], None).to(dev)
    x = ops.alloc_features(n, 100, torch.bfloat16, dev, pad_to=64)
    x.copy_(torch.randn(n, 100, device=dev).to(torch.bfloat16))
    labels = torch.randint(0, 47, (n,), device=dev)
    opt = FlatAdam(list(model.parameters()), lr=1e-3)          # as bench.py

    def step():
        opt.zero_grad(set_to_none=True)
        out = model.forward_graph(g, x)
        loss = ops.cross_entropy(out, labels, reduction="sum", fold_relu=True) / n
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step()
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    evs.sort(key=lambda e: e.time_range.start)
    t0 = evs[0].time_range.start
    tot = 0.0
    for e in evs:
        d = e.time_range.end - e.time_range.start
        tot += d
        if d >= 20:
            print("%9.1f us  +%8.1f us  %s" % (e.time_range.start - t0, d, e.name[:110] if e.name.startswith("void dgll") or e.name.startswith("dgll") else e.name[:330]))
    print("kernel time %.3f ms, span %.3f ms, %d kernels" % (tot / 1e3, (evs[-1].time_range.end - t0) / 1e3, len(evs)))
    small = {}
    for e in evs:
        d = e.time_range.end - e.time_range.start
        if d < 20:
            k = small.setdefault(e.name[:150], [0, 0.0])
            k[0] += 1; k[1] += d
    print("kernels under 20 us: %d, %.3f ms" % (sum(v[0] for v in small.values()), sum(v[1] for v in small.values()) / 1e3))
    for name, (cnt, us) in sorted(small.items(), key=lambda kv: -kv[1][1]):
        print("  %3d x %7.1f us  %s" % (cnt, us, name))


if __name__ == "__main__":
    main()
