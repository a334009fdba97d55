#!/usr/bin/env python3
"""BASELINE config 1 on the GPU: the reference's PPI evaluation loop (Evaluation/PPI/train_gcn.py:8-62) -- per-graph
full-batch steps of an L-layer GCN with Adam and CrossEntropyLoss over multi-hot labels -- on the HIP engine.

    python examples/ppi/train_gcn.py --data-dir /path/to/PPI            # GraphSAGE-format directory
    python examples/ppi/train_gcn.py --synthetic                        # PPI-shaped random graphs, no files needed

Prints one line per epoch like the reference (:53) and a final aggregated-edges/s figure."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from dgll_amd import ops  # noqa: E402
from dgll_amd.evaluation.ppi import GCN, load_ppi_dataset  # noqa: E402


def synthetic_split(n_graphs, seed, feats=50, classes=121):
    """PPI-shaped graphs: ~2.2 k nodes, ~28 directed edges per node, fp32 features, multi-hot labels."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n_graphs):
        n = int(torch.randint(1500, 3500, (1,), generator=g))
        e = torch.randint(0, n, (2, 14 * n), generator=g)
        e = torch.cat([e, e.flip(0)], dim=1)
        e = e[:, e[0] != e[1]]
        out.append((e, torch.randn(n, feats, generator=g), (torch.rand(n, classes, generator=g) < 0.3).float()))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data-dir", default=None)
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--lr", type=float, default=0.01)
    ap.add_argument("--capturable", action="store_true", help="eager loop with the capture-safe Adam (A/B against --hip-graphs)")
    ap.add_argument("--hip-graphs", action="store_true", help="capture each graph's training step into a HIP graph")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("this example runs the HIP kernels: a GPU is required")
    dev = torch.device("cuda:0")
    if args.synthetic or args.data_dir is None:
        train = synthetic_split(20, 0)
    else:
        train = load_ppi_dataset(args.data_dir, "train")
    train = [(e.to(dev), x.to(dev), y.to(dev)) for e, x, y in train]       # moved once, not per step (:37-39)
    torch.manual_seed(0)
    model = GCN(train[0][1].shape[1], args.hidden, train[0][2].shape[1], args.layers).to(dev)
    with torch.no_grad():
        for layer in model.layers:          # the reference's unscaled randn init overflows the loss on real PPI
            layer.weight.mul_(1.0 / layer.weight.shape[0] ** 0.5)
    opt = torch.optim.Adam(model.parameters(), lr=args.lr, capturable=args.hip_graphs or args.capturable)
    crit = ops.cross_entropy             # nn.CrossEntropyLoss()(out, float multi-hot labels) in one kernel per direction
    steps = None
    if args.hip_graphs:        # static shapes per graph: the whole epoch (one step per graph) becomes ONE HIP graph
        from dgll_amd.graphs import GraphedTrainStep

        t0 = time.time()
        steps = GraphedTrainStep([lambda e=e, x=x, y=y: crit(model(e, x), y) for e, x, y in train], opt, warmup=1)
        print("captured the %d steps of an epoch into one HIP graph in %.2f s (the warm-up epoch counts as training)"
              % (len(train), time.time() - t0))
    edges_per_epoch = sum(int(e.shape[1]) for e, _, _ in train) * args.layers * 2      # forward + transposed backward
    for epoch in range(args.epochs):
        torch.cuda.synchronize()
        t0 = time.time()
        total = torch.zeros((), device=dev)
        if steps is not None:
            steps()                                                      # one host call replays the epoch
            total = steps.total                                          # static scalar computed inside the graph
        else:
            for e, x, y in train:
                opt.zero_grad()
                loss = crit(model(e, x), y)
                loss.backward()
                opt.step()
                total += loss.detach()
        torch.cuda.synchronize()
        dt = time.time() - t0
        print("Epoch %d/%d, Loss: %.4f, Time per Epoch: %.4fs, %.1f M aggregated edges/s"
              % (epoch + 1, args.epochs, float(total), dt, edges_per_epoch / dt / 1e6))


if __name__ == "__main__":
    main()
