#!/usr/bin/env python3
"""Full-graph GraphSAGE training on the GPU (the reference's examples/graphsage/README.md "Full graph training"; its
train.py is a 7-line stub).  Synthetic products-shaped data by default -- there are no datasets offline:

    python examples/graphsage/train_full.py --nodes 200000 --epochs 20
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from dgll_amd import nn as dnn, ops, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=200_000)
    ap.add_argument("--avg-degree", type=int, default=50)
    ap.add_argument("--feats", type=int, default=100)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--classes", type=int, default=47)
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("this example runs the HIP kernels: a GPU is required")
    dev = torch.device("cuda:0")
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    torch.manual_seed(0)
    graph = synth.products_like_graph(dev, seed=0, n=args.nodes, n_undirected=args.nodes * args.avg_degree // 2, locality=0.9)
    n = graph.n_rows
    # labels that depend on the graph: the community a node was planted in, so the model has something to learn
    labels = ((torch.arange(n, device=dev) * 64 // n) % args.classes).long()      # the planted community (64 blocks)
    x = ops.alloc_features(n, args.feats, dtype, dev, pad_to=64)
    x.copy_(torch.randn(n, args.feats, device=dev) + torch.nn.functional.one_hot(labels % args.feats, args.feats) * 2.0)
    train = torch.rand(n, device=dev) < 0.1
    model = dnn.GraphSage(args.feats, [args.hidden, args.hidden, args.classes], None).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=3e-3)
    masked = torch.where(train, labels, torch.full_like(labels, -100))      # -100: ignored by the loss kernel
    for epoch in range(args.epochs):
        torch.cuda.synchronize()
        t0 = time.time()
        opt.zero_grad(set_to_none=True)
        out = model.forward_graph(graph, x)
        loss = ops.cross_entropy(out, masked)
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        dt = time.time() - t0
        with torch.no_grad():
            acc = float((out.argmax(1) == labels)[~train].float().mean())
        print("epoch %2d  loss %.4f  held-out acc %.3f  %.1f ms  %.2f G aggregated edges/s"
              % (epoch, float(loss), acc, dt * 1e3, 5 * graph.nnz / dt / 1e9))


if __name__ == "__main__":
    main()
