#!/usr/bin/env python3
"""Mini-batch GraphSAGE training on sampled blocks -- the loop of the reference's graphage.py:47-63 (DataLoader -> neighbour
sampler -> features of the sampled nodes -> model -> loss -> optimizer) on the GPU path of this package:

    sampler threads (native, per-batch seeds: every batch bit-equal to the reference loop under random.seed(batch_seed))
      -> loading stage (hot-node feature cache in HBM + pinned host rows, the outermost hop reduced straight out of the cache)
      -> consumer (hop-pyramid forward / loss / backward / FlatAdam; --hip-graph: replayed as one HIP graph on static shapes)

Synthetic community-structured data by default -- there are no datasets offline:

    python examples/graphsage/train_minibatch.py --nodes 100000 --epochs 3 --hip-graph
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from dgll_amd import nn as dnn, ops, synth  # noqa: E402
from dgll_amd.cache import GraphCacheServer  # noqa: E402
from dgll_amd.data import DGraph  # noqa: E402
from dgll_amd.dataloader import DataLoader  # noqa: E402
from dgll_amd.optim import FlatAdam  # noqa: E402
from dgll_amd.pipeline import MiniBatchPipeline  # noqa: E402
from dgll_amd.sampling import FastNeighborSampler  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=100_000)
    ap.add_argument("--avg-degree", type=int, default=60)
    ap.add_argument("--feats", type=int, default=128)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--classes", type=int, default=16)
    ap.add_argument("--fanouts", default="15,10,5")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--cache-frac", type=float, default=1.0, help="share of the nodes whose features are cached in HBM (by degree)")
    ap.add_argument("--sampler-threads", type=int, default=4)
    ap.add_argument("--hip-graph", action="store_true", help="replay forward + loss + backward as one HIP graph (graphs.GraphedSampledStep)")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("this example runs the HIP kernels: a GPU is required")
    dev = torch.device("cuda:0")
    fanouts = [int(f) for f in args.fanouts.split(",")]
    L = len(fanouts)
    torch.manual_seed(args.seed)
    g = synth.products_like_graph(dev, seed=args.seed, n=args.nodes, n_undirected=args.nodes * args.avg_degree // 2, locality=0.9,
                                  n_blocks=args.classes)
    n = g.n_rows
    indptr, indices = g.rowptr.cpu().numpy(), g.col.cpu().numpy().astype(np.int64)
    deg = g.degrees().cpu()
    del g
    labels = (torch.arange(n) * args.classes // n).long()                  # the planted community: something to learn
    feats = (torch.randn(n, args.feats) + torch.nn.functional.one_hot(labels % args.feats, args.feats) * 1.5).to(torch.bfloat16)
    dg = DGraph.from_csr(indptr, indices, labels=labels, features=feats)
    cache = GraphCacheServer(feats, gpuid=0)
    cache.auto_cache(deg, capacity=int(args.cache_frac * n))
    perm = torch.randperm(n)
    train, held_out = perm[: n // 2], perm[n // 2: n // 2 + 8 * args.batch]
    model = dnn.GraphSage(args.feats, [args.hidden] * (L - 1) + [args.classes], fanouts).to(dev)
    # the reference applies the activation after the LAST layer too (sageconv.py:83): logits >= 0, and a step that pushes a row's
    # logits below zero leaves it without gradient.  A classifier trains more robustly on unconstrained logits:
    model.gcn[-1].activation = None
    opt = FlatAdam(list(model.parameters()), lr=args.lr)
    device_graph = (torch.from_numpy(indptr).to(dev), torch.from_numpy(indices).to(dev))
    graphed, seen = None, [0] * L

    def pipeline(nodes, epoch):
        loader = DataLoader(dg, nodes, FastNeighborSampler(fanouts, defer_last_hop=True), batch_size=args.batch)
        return MiniBatchPipeline(loader, cache=cache, labels=labels, queue_size=4, device=dev, hops="sampled", reduce_last_hop="mean",
                                 sampler_threads=args.sampler_threads, base_seed=args.seed, epoch=epoch, device_graph=device_graph,
                                 build_blocks=True)

    def eager_step(b):
        out = model.forward_sampled(b.features, b.blocks, last_hop_reduced=b.last_hop_reduced)
        loss = ops.cross_entropy(out, b.labels)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    compute = torch.cuda.Stream(dev, priority=-1)          # the loading stage's kernels run beside the consumer's: give these priority
    compute.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(compute):
        for epoch in range(args.epochs):
            torch.cuda.synchronize()
            t0, total, steps = time.time(), 0.0, 0
            for b in pipeline(train[torch.randperm(train.numel())], epoch):
                if args.hip_graph and graphed is None and steps == 8:
                    # the first batches ran launch by launch and showed how far the hops fill; capture on those bounds + 10 %
                    from dgll_amd.graphs import GraphedSampledStep

                    rows = [args.batch] + [-(-int(r * 1.1) // 64) * 64 for r in seen[1:]]
                    graphed = GraphedSampledStep(model, opt, args.batch, fanouts, args.feats, args.classes, device=dev, rows=rows)
                loss = None
                if graphed is not None:
                    try:
                        loss = graphed(b)
                    except ValueError:                      # a batch beyond the captured bounds
                        loss = None
                if loss is None:
                    loss = eager_step(b)
                    for h in range(L):
                        seen[h] = max(seen[h], int(b.features[h].shape[0]))
                total += float(loss.detach()) if steps % 16 == 0 else 0.0
                steps += 1
            torch.cuda.synchronize()
            dt = time.time() - t0
            correct = count = 0
            with torch.no_grad():
                for b in pipeline(held_out, args.epochs + epoch):
                    out = model.forward_sampled(b.features, b.blocks, last_hop_reduced=b.last_hop_reduced)
                    correct += int((out.argmax(1) == b.labels).sum())
                    count += int(b.labels.numel())
            print("epoch %d  loss %.4f  held-out acc %.3f  %d batches in %.2f s = %.0f batches/s%s"
                  % (epoch, total / max(1, -(-steps // 16)), correct / max(count, 1), steps, dt, steps / dt,
                     "  (HIP graph)" if graphed is not None else ""), flush=True)


if __name__ == "__main__":
    main()
