#!/usr/bin/env python3
"""Where a mini-batch's period goes on the GPU (config 2 shape, the bench's own pipeline): kernel time per batch on the consumer's
stream and on the loading stream (torch profiler over 8 steady-state batches), next to the host time of both threads.
    python tools/minibatch_step_trace.py [--graph]"""
import os
import sys
import time
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from dgll_amd import nn as dnn, ops, synth  # noqa: E402
from dgll_amd.cache import GraphCacheServer  # noqa: E402
from dgll_amd.data import DGraph  # noqa: E402
from dgll_amd.dataloader import DataLoader  # noqa: E402
from dgll_amd.optim import FlatAdam  # noqa: E402
from dgll_amd.pipeline import MiniBatchPipeline  # noqa: E402
from dgll_amd.sampling import FastNeighborSampler  # noqa: E402


NAMEW = 220 if "--wide" in sys.argv else 90          # characters of a kernel's name kept


def main():
    dev = torch.device("cuda:0")
    n, f, classes, batch, fanouts = 232965, 602, 41, 1024, [25, 10, 10]
    g = synth.products_like_graph(dev, seed=1, n=n, n_undirected=57_300_000, locality=0.0, exact=True)
    indptr, indices = g.rowptr.cpu().numpy(), g.col.cpu().numpy().astype(np.int64)
    deg = g.degrees().cpu()
    del g
    torch.manual_seed(0)
    feats = torch.randn(n, f).to(torch.bfloat16)
    labels = torch.randint(0, classes, (n,))
    dg = DGraph.from_csr(indptr, indices, labels=labels, features=feats)
    cache = GraphCacheServer(feats, gpuid=0)
    cache.auto_cache(deg, capacity=n // 2)
    n_batches = 16 + 8 + 8
    train = torch.randperm(n)[:n_batches * batch]
    loader = DataLoader(dg, train, FastNeighborSampler(fanouts, defer_last_hop=True), batch_size=batch)
    device_graph = (torch.from_numpy(indptr).to(dev), torch.from_numpy(indices).to(dev))
    pipe = MiniBatchPipeline(loader, cache=cache, labels=labels, queue_size=4, device=dev, hops="sampled", reduce_last_hop="mean",
                             sampler_threads=8, base_seed=0, epoch=0, device_graph=device_graph, build_blocks=True)
    model = dnn.GraphSage(f, [256, 256, classes], fanouts).to(dev)
    opt = FlatAdam(list(model.parameters()), lr=1e-3)
    graphed = None
    if "--graph" in sys.argv:                 # the consumer's step as one HIP graph (dgll_amd.graphs.GraphedSampledStep)
        from dgll_amd.graphs import GraphedSampledStep

        graphed = GraphedSampledStep(model, opt, batch, fanouts, f, classes, device=dev)
    compute = torch.cuda.Stream(dev, priority=-1)
    compute.wait_stream(torch.cuda.current_stream(dev))
    prof = profile(activities=[ProfilerActivity.CUDA])
    done = 0
    with torch.cuda.stream(compute):
        for b in pipe:
            if done == 16:
                torch.cuda.synchronize()
                prof.__enter__()
                t0 = time.perf_counter()
            if done == 24:
                torch.cuda.synchronize()
                wall = time.perf_counter() - t0
                prof.__exit__(None, None, None)
            if graphed is not None:
                graphed(b)
            else:
                out = model.forward_sampled(b.features, b.blocks, last_hop_reduced=b.last_hop_reduced)
                loss = ops.cross_entropy(out, b.labels)
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
            done += 1
    torch.cuda.synchronize()
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    by_stream = {}
    for e in evs:
        by_stream.setdefault(getattr(e, "device_resource_id", getattr(e, "stream", 0)), []).append(e)
    print("8 batches in %.2f ms wall (%.2f ms per batch)" % (wall * 1e3, wall * 1e3 / 8))
    for sid, es in sorted(by_stream.items(), key=lambda kv: -len(kv[1])):
        tot = sum(e.time_range.end - e.time_range.start for e in es)
        print("stream %s: %d kernels / copies, %.3f ms busy per batch, %.1f launches per batch" % (sid, len(es), tot / 8e3, len(es) / 8))
        cnt, tim = Counter(), Counter()
        for e in es:
            cnt[e.name[:NAMEW]] += 1
            tim[e.name[:NAMEW]] += e.time_range.end - e.time_range.start
        for name, t in tim.most_common(22):
            print("    %5.1f x %7.1f us  %s" % (cnt[name] / 8, t / cnt[name], name))


if __name__ == "__main__":
    main()
