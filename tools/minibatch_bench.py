#!/usr/bin/env python3
"""BASELINE config 2 shape: 3-layer GraphSAGE on a Reddit-shaped synthetic graph, mini-batch training with neighbour
sampling (fan-out 25-10-10), hidden 256, bf16 -- the MQ-GNN style pipeline end to end:

  host: native bit-exact sampler (FastNeighborSampler)  ->  bounded queue (MiniBatchPipeline, side HIP stream)
  GPU : hot-node feature cache gather (GraphCacheServer) -> hop-pyramid GraphSage on CSR blocks -> loss/backward/Adam

Prints epoch time, batches/s and aggregated sampled edges/s.   python tools/minibatch_bench.py [--batches 20]
"""
import argparse
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dgll_amd import nn as dnn  # noqa: E402
from dgll_amd import synth  # noqa: E402
from dgll_amd.cache import GraphCacheServer  # noqa: E402
from dgll_amd.data import DGraph  # noqa: E402
from dgll_amd.dataloader import DataLoader  # noqa: E402
from dgll_amd.pipeline import MiniBatchPipeline  # noqa: E402
from dgll_amd.sampling import FastNeighborSampler  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=232_965)
ap.add_argument("--undirected-edges", type=int, default=57_300_000)   # ~114.6 M directed (Reddit)
ap.add_argument("--feats", type=int, default=602)
ap.add_argument("--classes", type=int, default=41)
ap.add_argument("--hidden", type=int, default=256)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--fanouts", default="25,10,10")
ap.add_argument("--train-nodes", type=int, default=153_431)
ap.add_argument("--batches", type=int, default=0, help="0 = a full epoch")
ap.add_argument("--cache-frac", type=float, default=1.0)
args = ap.parse_args()
dev = torch.device("cuda:0")
fanouts = [int(x) for x in args.fanouts.split(",")]

t0 = time.time()
g = synth.products_like_graph(dev, seed=1, n=args.nodes, n_undirected=args.undirected_edges, locality=0.0)
indptr, indices = g.rowptr.cpu().numpy(), g.col.cpu().numpy().astype(np.int64)
deg = g.degrees().cpu()
del g
torch.manual_seed(0)
feats = torch.randn(args.nodes, args.feats).to(torch.bfloat16)
labels = torch.randint(0, args.classes, (args.nodes,))
dg = DGraph.from_csr(indptr, indices, labels=labels, features=feats)
print("graph: %d nodes, %d directed edges, avg degree %.0f (built in %.1f s)" % (args.nodes, len(indices), len(indices) / args.nodes, time.time() - t0), flush=True)

cache = GraphCacheServer(feats, gpuid=0)
cache.log = True
cache.auto_cache(deg, capacity=int(args.cache_frac * args.nodes))
train = torch.randperm(args.nodes)[:args.train_nodes]
if args.batches:
    train = train[:args.batches * args.batch]
class TimedSampler(FastNeighborSampler):
    seconds = 0.0

    def sample(self, g_, seeds):
        t = time.perf_counter()
        out = super().sample(g_, seeds)
        TimedSampler.seconds += time.perf_counter() - t
        return out


loader = DataLoader(dg, train, TimedSampler(fanouts, defer_last_hop=True), batch_size=args.batch)
L = len(fanouts)


def hop_ids(b):          # hop 0 = seeds, hop h+1 = sources sampled around hop h (subgs are outermost first)
    return [b.output_nodes] + [b.subgraphs[L - 1 - h].src_nodes() for h in range(L)]


pipe = MiniBatchPipeline(loader, cache=cache, labels=labels, queue_size=4, device=dev, hops=hop_ids)
_fetch, fetch_s = pipe._fetch, [0.0]


def timed_fetch(ids):
    t = time.perf_counter()
    out = _fetch(ids)
    fetch_s[0] += time.perf_counter() - t
    return out


pipe._fetch = timed_fetch
_hops, hops_s = pipe.hops, [0.0]


def timed_hops(b):
    t = time.perf_counter()
    out = _hops(b)
    hops_s[0] += time.perf_counter() - t
    return out


pipe.hops = timed_hops
_put, put_s = pipe.queue.put, [0.0]


def timed_put(item):
    t = time.perf_counter()
    _put(item)
    put_s[0] += time.perf_counter() - t


pipe.queue.put = timed_put
blocks_s = [0.0]
model = dnn.GraphSage(args.feats, [args.hidden] * (L - 1) + [args.classes], fanouts).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
random.seed(0)
torch.cuda.synchronize()
t0 = time.time()
n_batches = n_edges = 0
gpu_ms = 0.0
for b in pipe:
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    tb = time.perf_counter()
    blocks = [b.subgraphs[L - 1 - h].to_block(dev) for h in range(L)]
    blocks_s[0] += time.perf_counter() - tb
    out = model.forward_sampled(b.features, blocks)
    loss = torch.nn.functional.cross_entropy(out.float(), b.labels)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    ev1.record()
    n_batches += 1
    if n_batches > 2:
        ev1.synchronize()
        gpu_ms += ev0.elapsed_time(ev1)
    # edges aggregated per batch: layer l runs over hops 0..L-l-1, each an SpMM over that hop's sampled edges (fwd + bwd)
    per_hop = [b.subgraphs[L - 1 - h].num_src_nodes() for h in range(L)]
    n_edges += sum(sum(per_hop[:L - l]) for l in range(L)) * 2
torch.cuda.synchronize()
dt = time.time() - t0
print("%d batches of %d seeds in %.2f s: %.1f ms/batch, %.2f M aggregated sampled edges/s, final loss %.3f, cache miss rate %.3f" % (
    n_batches, args.batch, dt, dt / n_batches * 1e3, n_edges / dt / 1e6, float(loss), cache.get_miss_rate()), flush=True)
print("producer-side feature fetch calls (host time): %.1f ms/batch" % (fetch_s[0] / n_batches * 1e3))
print("host sampler: %.1f ms/batch; GPU side (blocks + forward/backward/Adam, batches 3..): %.1f ms/batch" % (
    TimedSampler.seconds / n_batches * 1e3, gpu_ms / max(n_batches - 2, 1)), flush=True)
print("producer: hop-id lists %.1f ms/batch, blocked on a full queue %.1f ms/batch; consumer: block build host time %.1f ms/batch" % (
    hops_s[0] / n_batches * 1e3, put_s[0] / n_batches * 1e3, blocks_s[0] / n_batches * 1e3))
if not args.batches:
    print("epoch time (%d train nodes): %.2f s" % (args.train_nodes, dt), flush=True)
    # a second epoch: nothing left to warm up (library handles, GEMM heuristics, allocator pools)
    torch.cuda.synchronize()
    t1 = time.time()
    for b in pipe:
        blocks = [b.subgraphs[L - 1 - h].to_block(dev) for h in range(L)]
        out = model.forward_sampled(b.features, blocks)
        loss = torch.nn.functional.cross_entropy(out.float(), b.labels)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    print("second epoch: %.2f s" % (time.time() - t1), flush=True)
