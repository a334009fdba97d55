"""Named profiler ranges (dgll_amd/ranges.py): the reference's record_function names (FeatureCache/gs.py:88,93;
storage.py:164-195) behind one switch, a shared no-op when it is off."""
import random

import torch

from dgll_amd import ranges


def test_off_is_one_shared_noop_and_on_is_record_function():
    ranges.enable(False)
    try:
        a, b = ranges.rng("gpu-load"), ranges.rng("sample")
        assert a is b                                   # no allocation per call when the switch is off
        with a:
            pass
        ranges.enable(True)
        assert isinstance(ranges.rng("gpu-load"), torch.profiler.record_function)
    finally:
        ranges.enable(False)
    assert set(("sample", "gpu-load", "cache-index", "cache-gpu", "cache-cpu", "consume", "exchange", "racom-allreduce")) == set(ranges.NAMES)


def test_pipeline_stages_show_up_under_the_profiler_on_host_tensors():
    """sample / gpu-load around the host pipeline (no GPU: features come from Dgraph.get_features), racom-allreduce around a
    world-size-1 reduction: the ranges appear in torch.profiler's table with one call per batch."""
    from dgll_amd import dist as ddist
    from dgll_amd.data import DGraph
    from dgll_amd.dataloader import DataLoader
    from dgll_amd.pipeline import MiniBatchPipeline
    from dgll_amd.sampling import FastNeighborSampler

    n = 600
    rng_ = random.Random(0)
    edges = [[rng_.randrange(n) for _ in range(rng_.randrange(1, 12))] for _ in range(n)]
    dg = DGraph(nodes=torch.arange(n), edges=edges, labels=torch.arange(n) % 3, features=torch.randn(n, 4))
    ranges.enable(True)
    try:
        with ranges.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
            for threads in (0, 2):
                loader = DataLoader(dg, torch.arange(320), FastNeighborSampler([4, 3]), batch_size=64)
                pipe = MiniBatchPipeline(loader, labels=dg.labels, queue_size=2, device="cpu", hops="sampled", sampler_threads=threads)
                assert [b.step for b in pipe] == list(range(5))
            w = torch.nn.Parameter(torch.ones(3))
            w.grad = torch.ones(3)
            ddist.RaCoM([w], "cpu").all_reduce_and_wait()
    finally:
        ranges.enable(False)
    counts = {e.key: e.count for e in prof.key_averages() if e.key in ranges.NAMES}
    assert counts.get("gpu-load") == 10 and counts.get("sample", 0) >= 10 and counts.get("racom-allreduce") == 2
