#!/usr/bin/env python3
"""2-layer sparse GAT (8 heads) on the GPU -- BASELINE config 4's model (the reference's examples/gat/train.py is empty;
its SpGAT needs a dense N x N adjacency, gatconv.py:115, and cannot run at this size).  Synthetic data:

    python examples/gat/train.py --nodes 200000 --epochs 10
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
    ap.add_argument("--hidden", type=int, default=32, help="per head")
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--classes", type=int, default=47)
    ap.add_argument("--epochs", type=int, default=10)
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("this example runs the HIP kernels: a GPU is required")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    graph = synth.products_like_graph(dev, seed=0, n=args.nodes, n_undirected=args.nodes * args.avg_degree // 2, locality=0.9,
                                      self_loops=True)                     # every row needs an edge (gatconv.py:139-141)
    n = graph.n_rows
    labels = ((torch.arange(n, device=dev) * 64 // n) % args.classes).long()      # the planted community (64 blocks)
    x = (torch.randn(n, args.feats, device=dev) + torch.nn.functional.one_hot(labels % args.feats, args.feats) * 2.0).to(torch.bfloat16)
    model = dnn.SpGAT(args.feats, args.hidden, args.classes, dropout=0.0, alpha=0.2, nheads=args.heads).to(dev).to(torch.bfloat16)
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    for epoch in range(args.epochs):
        torch.cuda.synchronize()
        t0 = time.time()
        opt.zero_grad(set_to_none=True)
        logp = model(x, graph)                                  # log_softmax output, as the reference's SpGAT.forward
        loss = torch.nn.functional.nll_loss(logp.float(), labels)
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        dt = time.time() - t0
        acc = float((logp.argmax(1) == labels).float().mean())
        print("epoch %2d  loss %.4f  acc %.3f  %.1f ms" % (epoch, float(loss), acc, dt * 1e3))


if __name__ == "__main__":
    main()
