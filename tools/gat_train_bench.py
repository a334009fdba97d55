#!/usr/bin/env python3
"""BASELINE config 4 shape on one GPU: 2-layer SpGAT (8 heads x 32 -> 47 classes) full-graph training step on the
products-shaped graph, bf16 activations.  Prints step time and the per-kernel launch table."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import nn as dnn  # noqa: E402
from dgll_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
loc = float(sys.argv[1]) if len(sys.argv) > 1 else 0.9
reorder = (sys.argv[2] if len(sys.argv) > 2 else "lpa")
g = synth.products_like_graph(dev, seed=0, locality=loc, self_loops=True, exact=True, permute_ids=True)   # bench.py's graph + I
if reorder != "none":
    g = g.reorder(method=reorder, seed=0)[0]
n = g.n_rows
torch.manual_seed(0)
model = dnn.SpGAT(100, 32, 47, dropout=0.0, alpha=0.2, nheads=8).to(dev)
x = ops.alloc_features(n, 100, torch.bfloat16, dev, pad_to=64)
x.copy_(torch.randn(n, 100, device=dev))
labels = torch.randint(0, 47, (n,), device=dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)


def step():
    opt.zero_grad(set_to_none=True)
    out = model(x, g)                       # log_softmax output
    loss = -out.float().gather(1, labels.unsqueeze(1)).sum() / n
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 10
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
print("SpGAT 100 -> 8x32 -> 47, products-shaped (locality %.1f, nnz %d): %.1f ms/step (fwd+bwd+Adam), loss %.3f" % (loc, g.nnz, ms, float(loss.detach())))
