#!/usr/bin/env python3
"""Per-step global loss of the partitioned full-graph GraphSAGE training loop (run under torch.distributed.run with
DGLL_BENCH_BACKEND=gloo to share one GPU); the 1-rank and N-rank traces must agree step by step."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dgll_amd import dist as ddist, nn as dnn, ops, synth  # noqa: E402

world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
if world > 1:
    dist.init_process_group(os.environ.get("DGLL_BENCH_BACKEND", "nccl"))
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
torch.manual_seed(0)
full = synth.products_like_graph(dev, seed=0, n=nodes, n_undirected=nodes * 25, locality=0.9)
n = full.n_rows
gen = torch.Generator(device=dev); gen.manual_seed(1)
model = dnn.GraphSage(100, [256, 256, 47], None).to(dev)
labels_all = torch.randint(0, 47, (n,), generator=gen, device=dev)
feats = torch.randn(n, 100, generator=gen, device=dev)
part = ddist.partition_contiguous(full, world, rank)
engine = ddist.DistGraph(part, dev); engine.verify()
x_local = ops.alloc_features(part.n_own, 100, torch.bfloat16, dev, pad_to=64)
x_local.copy_(engine.permute_to_local(feats[part.own_begin:part.own_end]).to(torch.bfloat16))
labels = engine.permute_to_local(labels_all[part.own_begin:part.own_end])
placed = engine.place_input_halo(x_local)
racom = ddist.RaCoM(model.parameters(), dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
for it in range(steps):
    opt.zero_grad(set_to_none=True)
    out = engine.sage_forward(model, x_local, placed)
    logp = torch.log_softmax(out.float(), dim=1)
    loss = -logp.gather(1, labels.unsqueeze(1)).sum() * (world / n)
    loss.backward()
    racom.all_reduce_and_wait()
    opt.step()
    g = loss.detach().double() / world
    fin = torch.tensor([float(torch.isfinite(out).all())], device=dev)
    if world > 1:
        dist.all_reduce(g); dist.all_reduce(fin, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("step %d loss %.6f finite_out %d" % (it, float(g), int(fin)), flush=True)
if world > 1:
    dist.destroy_process_group()
