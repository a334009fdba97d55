#!/usr/bin/env python3
"""The per-range table the reference prints at the end of a run (`prof.key_averages().table(sort_by='cuda_time_total')`,
FeatureCache/gs.py:82,113), for this engine's pipeline: a small sampled-GraphSAGE run (bit-exact sampler threads -> loading stage
with the hot-node cache -> training) under torch.profiler with DGLL_PROFILE_RANGES on, reduced to the named ranges of
dgll_amd/ranges.py.

    python tools/range_table.py [--batches 40] [--nodes 200000] [--all]      (--all: every profiler row, not only the named ranges)
"""
import argparse
import os
import sys

os.environ["DGLL_PROFILE_RANGES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dgll_amd import nn as dnn, ops, ranges, synth  # noqa: E402
from dgll_amd.cache import GraphCacheServer  # noqa: E402
from dgll_amd.data import DGraph  # noqa: E402
from dgll_amd.dataloader import DataLoader  # noqa: E402
from dgll_amd.optim import FlatAdam  # noqa: E402
from dgll_amd.pipeline import MiniBatchPipeline  # noqa: E402
from dgll_amd.sampling import FastNeighborSampler  # noqa: E402


def run(n_batches, nodes, feats=128, classes=16, batch=256, fanouts=(10, 5, 5), device="cuda"):
    dev = torch.device(device)
    g = synth.products_like_graph(dev, seed=1, n=nodes, n_undirected=nodes * 20, locality=0.0, exact=True)
    indptr, indices = g.rowptr.cpu().numpy(), g.col.cpu().numpy().astype(np.int64)
    deg = g.degrees().cpu()
    x = torch.randn(nodes, feats).to(torch.bfloat16)
    labels = torch.randint(0, classes, (nodes,))
    dg = DGraph.from_csr(indptr, indices, labels=labels, features=x)
    cache = GraphCacheServer(x, gpuid=dev.index or 0)
    cache.auto_cache(deg, capacity=nodes // 2)
    train = torch.randperm(nodes)[:n_batches * batch]
    loader = DataLoader(dg, train, FastNeighborSampler(list(fanouts), defer_last_hop=True), batch_size=batch)
    pipe = MiniBatchPipeline(loader, cache=cache, labels=labels, queue_size=4, device=dev, hops="sampled", reduce_last_hop="mean",
                             sampler_threads=2, device_graph=(torch.from_numpy(indptr).to(dev), torch.from_numpy(indices).to(dev)),
                             build_blocks=True)
    model = dnn.GraphSage(feats, [64, 64, classes], list(fanouts)).to(dev)
    opt = FlatAdam(list(model.parameters()), lr=1e-3)
    for b in pipe:
        with ranges.rng("consume"):
            out = model.forward_sampled(b.features, b.blocks, last_hop_reduced=b.last_hop_reduced)
            loss = ops.cross_entropy(out, b.labels)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
    torch.cuda.synchronize()
    return float(loss)


def table(prof, only_named=True):
    rows = [e for e in prof.key_averages() if (e.key in ranges.NAMES or not only_named)]
    rows.sort(key=lambda e: -getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0)))
    lines = ["%-18s %8s %14s %14s %14s" % ("range", "calls", "CPU total ms", "GPU total ms", "CPU ms / call")]
    for e in rows:
        dev_t = getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0))
        lines.append("%-18s %8d %14.3f %14.3f %14.4f" % (e.key[:18], e.count, e.cpu_time_total / 1e3, dev_t / 1e3, e.cpu_time_total / 1e3 / max(e.count, 1)))
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=40)
    ap.add_argument("--nodes", type=int, default=200_000)
    ap.add_argument("--all", action="store_true")
    args = ap.parse_args()
    run(4, args.nodes)                       # warm-up: library load, plans, allocator
    acts = [torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]
    with ranges.profile(activities=acts) as prof:
        run(args.batches, args.nodes)
    print(table(prof, only_named=not args.all))


if __name__ == "__main__":
    main()
