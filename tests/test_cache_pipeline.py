"""f2: hot-node feature cache (one-launch hit/miss gather) and the bounded mini-batch queues."""
import random
import sys
import threading
import time

import numpy as np
import pytest
import torch


def test_pipeline_queue_is_bounded_and_ordered_on_cpu():
    """Producer/consumer semantics without a GPU: batches arrive in order, the producer never runs more than
    `queue_size` (+1 in hand) ahead, errors surface in the consumer."""
    from dgll_amd.pipeline import MiniBatchPipeline

    produced = []

    class FakeGraph:
        def get_features(self, ids):
            return torch.zeros(len(ids), 2)

    class FakeLoader:
        Dgraph = FakeGraph()

        def __iter__(self):
            for i in range(12):
                produced.append(i)
                yield torch.tensor([i]), torch.tensor([i]), []

    pipe = MiniBatchPipeline(FakeLoader(), queue_size=3, device="cpu")
    seen = []
    for b in pipe:
        time.sleep(0.02)
        assert len(produced) - len(seen) <= 3 + 2 + 4      # loaded queue + consumer/loader hands + sampled hand-over (2) + sampler hand
        seen.append(b.step)
    assert seen == list(range(12))

    class Broken(FakeLoader):
        def __iter__(self):
            yield torch.tensor([0]), torch.tensor([0]), []
            raise RuntimeError("sampler failed")

    with pytest.raises(RuntimeError, match="sampler failed"):
        list(MiniBatchPipeline(Broken(), queue_size=2, device="cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,dim", [(torch.float32, 100), (torch.float32, 7), (torch.bfloat16, 256), (torch.bfloat16, 50)])
def test_fetch_data_equals_indexing(cuda_device, dtype, dim):
    """(node-id batch, cached-id set) -> gathered features == features[ids]; miss accounting (storage.py:213-220)."""
    from dgll_amd.cache import GraphCacheServer, gather_rows

    n = 5000
    torch.manual_seed(1)
    feats = torch.randn(n, dim).to(dtype)
    deg = torch.randint(0, 1000, (n,))
    srv = GraphCacheServer(feats, gpuid=0)
    srv.log = True
    srv.auto_cache(deg, capacity=1200)                     # partial: top-1200 out-degree nodes
    assert srv.cached_num == 1200 and not srv.full_cached
    top = set(torch.argsort(deg, descending=True, stable=True)[:1200].tolist())
    ids = torch.randint(0, n, (3000,), device=cuda_device)
    got = srv.fetch_data(ids)
    assert torch.equal(got.cpu(), feats[ids.cpu()])
    expect_miss = sum(1 for v in ids.tolist() if v not in top)
    assert abs(srv.get_miss_rate() - expect_miss / 3000) < 1e-9
    srv.auto_cache(deg, capacity=n)                        # everything fits: full cache, no misses
    assert srv.full_cached
    assert torch.equal(srv.fetch_data(ids).cpu(), feats[ids.cpu()])
    assert srv.get_miss_rate() == 0.0
    dev_feats = feats.to(cuda_device)
    assert torch.equal(gather_rows(dev_feats, ids), dev_feats[ids])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,dim,reduce", [(torch.bfloat16, 602, "mean"), (torch.bfloat16, 256, "sum"), (torch.float32, 100, "mean"),
                                              (torch.float32, 7, "mean"), (torch.bfloat16, 50, "mean"), (torch.bfloat16, 1400, "mean")])
def test_aggregate_data_equals_fetch_then_reduce(cuda_device, dtype, dim, reduce):
    """GraphCacheServer.aggregate_data: the neighbour reduction read straight from the HBM cache / the pinned host rows ==
    fetch_data followed by the per-row mean (sageconv.py:33-36) -- ragged rows, empty rows, hits and misses mixed inside a row,
    rows wider than one column block; the miss accounting counts every id once."""
    from dgll_amd.cache import GraphCacheServer

    n = 4000
    torch.manual_seed(dim)
    feats = torch.randn(n, dim).to(dtype)
    deg = torch.randint(0, 1000, (n,))
    srv = GraphCacheServer(feats, gpuid=0)
    srv.log = True
    srv.auto_cache(deg, capacity=1000)
    top = set(torch.argsort(deg, descending=True, stable=True)[:1000].tolist())
    counts = torch.randint(0, 30, (700,))
    counts[::50] = 0                                                       # rows without neighbours
    counts[1] = 257                                                        # a long row
    rowptr = torch.zeros(701, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(counts, 0)
    ids = torch.randint(0, n, (int(rowptr[-1]),))
    got = srv.aggregate_data(ids.to(cuda_device), rowptr, reduce=reduce)
    assert got.shape == (700, dim) and got.dtype == dtype and (got.stride(0) * got.element_size()) % 16 == 0
    rows = torch.repeat_interleave(torch.arange(700), counts)
    ref = torch.zeros(700, dim, dtype=torch.float64).index_add_(0, rows, feats[ids].double())
    if reduce == "mean":
        ref = ref / counts.clamp(min=1).double()[:, None]
    tol = 2.0 ** -8 if dtype == torch.bfloat16 else 1e-6
    err = (got.double().cpu() - ref).abs()
    assert float((err - tol * ref.abs()).max()) <= (2e-2 if dtype == torch.bfloat16 else 1e-5), float(err.max())
    assert bool((got[::50] == 0).all())                                    # empty rows: zeros
    expect_miss = sum(1 for v in ids.tolist() if v not in top)
    assert abs(srv.get_miss_rate() - expect_miss / ids.numel()) < 1e-9
    srv.auto_cache(deg, capacity=n)                                        # full cache: same result, no miss
    again = srv.aggregate_data(ids.to(cuda_device), rowptr, reduce=reduce)
    assert torch.equal(again, got) and srv.get_miss_rate() == 0.0


@pytest.mark.gpu
def test_aggregate_data_refuses_rows_that_are_not_4_byte_granular(cuda_device):
    from dgll_amd.cache import GraphCacheServer

    srv = GraphCacheServer(torch.randn(100, 5).to(torch.bfloat16), gpuid=0)
    with pytest.raises(ValueError):
        srv.aggregate_data(torch.arange(10, device=cuda_device), torch.tensor([0, 4, 10]))


@pytest.mark.gpu
def test_cache_with_nid_map_and_pipeline_on_gpu(cuda_device):
    from conftest import load_golden
    from dgll_amd.cache import GraphCacheServer
    from dgll_amd.data import DGraph
    from dgll_amd.dataloader import DataLoader
    from dgll_amd.pipeline import MiniBatchPipeline
    from dgll_amd.sampling import DGLLNeighborSampler

    g = load_golden("sampler_n400")
    ptr, idx = g["adj_ptr"], g["adj_idx"]
    n = len(ptr) - 1
    edges = [idx[ptr[i]:ptr[i + 1]].tolist() for i in range(n)]
    feats = torch.randn(n, 32)
    labels = torch.arange(n) % 7
    dg = DGraph(nodes=torch.arange(n), edges=edges, labels=labels, features=feats)
    srv = GraphCacheServer(feats, gpuid=0)
    srv.auto_cache(torch.tensor([len(e) for e in edges]), capacity=100)
    loader = DataLoader(dg, torch.arange(0, 200), DGLLNeighborSampler([4, 3]), batch_size=64)
    random.seed(5)
    pipe = MiniBatchPipeline(loader, cache=srv, labels=labels, queue_size=2, device=cuda_device)
    steps = 0
    for b in pipe:
        assert torch.equal(b.features[0].cpu(), feats[b.input_nodes])
        assert torch.equal(b.labels.cpu(), labels[b.output_nodes])
        steps += 1
    assert steps == 4
    # local -> full id map (storage.py:27): features live under permuted ids on the host
    perm = torch.randperm(n)
    srv2 = GraphCacheServer(feats, nid_map=perm, gpuid=0)
    srv2.auto_cache(torch.arange(n), capacity=50)
    ids = torch.randint(0, n, (500,), device=cuda_device)
    assert torch.equal(srv2.fetch_data(ids).cpu(), feats[perm[ids.cpu()]])


@pytest.mark.gpu
def test_cache_refresh_during_iteration_never_serves_a_wrong_row(cuda_device):
    """The docstrings invite re-drawing the cached set periodically; the pipeline's loading thread fetches on its own stream
    at the same time.  Every fetched row must equal features[id] whatever pair (slot map, cache block) the fetch snapshot
    (published atomically, old pair kept alive by record_stream), the access counters survive the cross-stream hand-over,
    and the miss counters are read after the stream that wrote them."""
    from dgll_amd.cache import GraphCacheServer

    n, dim = 20000, 64
    torch.manual_seed(3)
    feats = torch.randn(n, dim).to(torch.bfloat16)
    feats[:, 0] = (torch.arange(n) % 251).to(torch.bfloat16)      # a column that identifies the row exactly
    srv = GraphCacheServer(feats, gpuid=0)
    srv.log = True
    srv.global_sampling_cache(torch.rand(n) + 0.1, capacity=4000, seed=1)
    load_stream = torch.cuda.Stream(device=cuda_device)
    stop, errors, fetched = threading.Event(), [], [0]

    def loader():
        gen = torch.Generator().manual_seed(7)
        try:
            while not stop.is_set():
                ids = torch.randint(0, n, (4096,), generator=gen)
                out = srv.fetch_data(ids, stream=load_stream)
                srv.record_access(ids, stream=load_stream)
                load_stream.synchronize()
                if not torch.equal(out.cpu(), feats[ids]):
                    errors.append("wrong rows after %d fetches" % fetched[0])
                    return
                fetched[0] += 1
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    t = threading.Thread(target=loader)
    t.start()
    for i in range(30):
        if i % 2:
            srv.refresh()
        else:
            srv.refresh_from_access()
        time.sleep(0.01)
    stop.set()
    t.join()
    assert not errors, errors
    assert fetched[0] > 5
    rate = srv.get_miss_rate()
    assert 0.5 < rate < 0.95           # 4000 of 20000 nodes cached: most fetches miss, none is lost from the count


def test_adaptive_queue_grows_when_the_consumer_starves_and_shrinks_when_the_producer_idles():
    """README.md:29 "adaptive queue-sizing strategy to balance computation and memory efficiency"."""
    from dgll_amd.pipeline import AdaptiveQueue

    # (a) bursty producer (a slow item every few), fast consumer -> the consumer starves -> the bound doubles, up to the cap
    q = AdaptiveQueue(size=2, min_size=2, max_size=16, window=4)

    def produce(n, slow_every, slow, fast):
        for i in range(n):
            time.sleep(slow if i % slow_every == 0 else fast)
            q.put(i)
        q.put(None)

    t = threading.Thread(target=produce, args=(40, 5, 0.03, 0.0))
    t.start()
    got = []
    while True:
        item = q.get()
        if item is None:
            break
        got.append(item)
        time.sleep(0.004)
    t.join()
    assert got == list(range(40))                       # FIFO, nothing lost
    assert max(q.history) > 2 and max(q.history) <= 16

    # (b) fast producer, slow consumer -> never starved, producer blocked most of the time -> shrinks towards min_size
    q = AdaptiveQueue(size=8, min_size=2, max_size=16, window=4)
    t = threading.Thread(target=produce, args=(48, 10 ** 9, 0.0, 0.0))
    t.start()
    n = 0
    while q.get() is not None:
        n += 1
        time.sleep(0.003)
    t.join()
    assert n == 48 and q.history[-1] < 8 and min(q.history) >= 2

    # (c) the memory bound caps the growth: 10 batches of this size fit the budget
    q = AdaptiveQueue(size=4, max_size=64)
    q.set_memory_bound(batch_bytes=100, budget_bytes=1000)
    assert q.max_size == 10
    q.set_memory_bound(batch_bytes=1000, budget_bytes=1500)      # not even two fit: the floor is min_size
    assert q.max_size == 2 and q.size == 2

    # (d) a fixed bound never moves
    q = AdaptiveQueue(size=3, adaptive=False, window=2)
    for i in range(3):
        q.put(i)
    assert [q.get() for _ in range(3)] == [0, 1, 2] and q.history == [3]


@pytest.mark.gpu
def test_global_sampling_cache_and_access_driven_refresh(cuda_device):
    """README.md:27-29 "global neighbor sampling with caching": the cache holds a weighted global sample of the nodes
    (re-drawn by refresh()), or the nodes fetched most often; whatever it holds, fetch_data == features[ids]."""
    from dgll_amd.cache import GraphCacheServer

    n, dim, cap = 20000, 64, 2000
    torch.manual_seed(2)
    feats = torch.randn(n, dim).to(torch.bfloat16)
    deg = torch.cat([torch.full((200,), 5000), torch.randint(1, 30, (n - 200,))])      # 200 hubs, then a light tail
    srv = GraphCacheServer(feats, gpuid=0)
    srv.log = True
    srv.global_sampling_cache(deg, capacity=cap, seed=1)
    assert srv.cached_num == cap and not srv.full_cached
    flag = srv.gpu_flag.cpu()
    assert int(flag[:200].sum()) >= 195                      # a hub is ~400x likelier than a tail node: (almost) all are in
    assert int(flag[200:].sum()) >= cap - 200                # the rest of the capacity is a sample of the tail
    ids = torch.randint(0, n, (5000,))
    assert torch.equal(srv.fetch_data(ids.cuda()).cpu(), feats[ids])
    first = flag.clone()
    srv.refresh()                                            # a new draw: the tail rotates, hubs stay
    second = srv.gpu_flag.cpu()
    assert int((first & second)[:200].sum()) >= 190 and int((first ^ second).sum()) > 1000
    assert torch.equal(srv.fetch_data(ids.cuda()).cpu(), feats[ids])
    # access-driven refresh: nodes 10000..10999 are fetched over and over -> they all move into the cache
    hot = torch.arange(10000, 11000)
    for _ in range(3):
        srv.record_access(hot.cuda())
    srv.record_access(ids.cuda())
    srv.refresh_from_access()
    assert bool(srv.gpu_flag[hot.cuda()].all())
    srv.get_miss_rate()
    got = srv.fetch_data(hot.cuda())
    assert torch.equal(got.cpu(), feats[hot]) and srv.get_miss_rate() == 0.0
    # capacity >= node count: everything cached
    srv.global_sampling_cache(deg, capacity=n)
    assert srv.full_cached


def _zipf_graph(n, seed):
    rng = np.random.default_rng(seed)
    edges = []
    for v in range(n):
        deg = int(min(n - 1, rng.zipf(1.3))) if v % 9 else 0
        edges.append(rng.choice(n, size=deg, replace=False).tolist())
    return edges


def test_per_batch_seeded_pipeline_with_sampler_threads_matches_the_reference_loop_batch_by_batch():
    """MiniBatchPipeline(sampler_threads=K): K native sampler threads each own whole batches, batch b of epoch e is drawn under
    batch_seed(base, e, b); batches arrive IN ORDER and every one is bit-equal to oracle/sampler.py (the reference's loop) run
    right after random.seed(that seed).  Host tensors, no cache: features come from Dgraph.get_features per hop."""
    from dgll_amd.data import DGraph
    from dgll_amd.dataloader import DataLoader
    from dgll_amd.pipeline import MiniBatchPipeline
    from dgll_amd.sampling import FastNeighborSampler
    from dgll_amd.sampling.fast_sampler import batch_seed
    from oracle import sampler as osampler

    n = 1500
    edges = _zipf_graph(n, 3)
    feats = torch.arange(n * 2, dtype=torch.float32).view(n, 2)
    dg = DGraph(nodes=torch.arange(n), edges=edges, labels=torch.arange(n) % 5, features=feats)
    fanouts = [5, 3]
    train = torch.randperm(n, generator=torch.Generator().manual_seed(0))[:700]
    for defer in (False, True):
        loader = DataLoader(dg, train, FastNeighborSampler(fanouts, defer_last_hop=defer), batch_size=64)
        pipe = MiniBatchPipeline(loader, labels=dg.labels, queue_size=3, device="cpu", hops="sampled", sampler_threads=4,
                                 base_seed=11, epoch=2)
        random.seed(1234)
        before = random.getstate()
        steps = []
        for b in pipe:
            i = b.step
            seeds = train[i * 64:(i + 1) * 64]
            random.seed(batch_seed(11, 2, i))
            inp, outp, layers = osampler.sample(edges, seeds.tolist(), fanouts)
            assert b.output_nodes.tolist() == outp and torch.as_tensor(b.input_nodes).tolist() == inp
            for sg, (src, dst) in zip(b.subgraphs, layers):
                assert sg.src_nodes().tolist() == src and sg.dst_nodes().tolist() == dst
            # hop pyramid of features: hop 0 = seeds, hop 1 = sources around them, hop 2 = the outermost sources
            assert len(b.features) == 3 and torch.equal(b.features[0], feats[seeds]) and torch.equal(b.features[2], feats[torch.tensor(inp)])
            assert torch.equal(b.labels, dg.labels[seeds])
            steps.append(i)
            random.seed(1234)          # whatever the consumer does with the global generator, the batches do not depend on it
        assert steps == list(range(len(loader)))
    random.setstate(before)


def test_200_fresh_threaded_pipelines_on_a_list_of_lists_dgraph_are_bit_equal_to_the_reference_loop():
    """Regression for the round-4 race: FastNeighborSampler._csr used to publish a zero-filled indptr before filling it, and K
    sampler threads calling it at once on the reference's own DGraph format (list-of-lists, dgraph.py:18-47) could sample a graph
    without edges (input_nodes == []).  200 FRESH samplers (each one builds its CSR copy under contention) x 4 threads; every
    batch equals oracle/sampler.py (base_sampler.py:45-58, dgllsampler.py:10-21) under that batch's seed."""
    from dgll_amd.data import DGraph
    from dgll_amd.dataloader import DataLoader
    from dgll_amd.pipeline import MiniBatchPipeline
    from dgll_amd.sampling import FastNeighborSampler
    from dgll_amd.sampling.fast_sampler import batch_seed
    from oracle import sampler as osampler

    n = 1500
    edges = _zipf_graph(n, 3)
    feats = torch.arange(n * 2, dtype=torch.float32).view(n, 2)
    dg = DGraph(nodes=torch.arange(n), edges=edges, labels=torch.arange(n) % 5, features=feats)
    fanouts = [5, 3]
    train = torch.randperm(n, generator=torch.Generator().manual_seed(0))[:512]
    before = random.getstate()
    expected = []
    for i in range(8):
        random.seed(batch_seed(11, 2, i))
        expected.append(osampler.sample(edges, train[i * 64:(i + 1) * 64].tolist(), fanouts))
    random.setstate(before)
    wrong = 0
    interval = sys.getswitchinterval()
    sys.setswitchinterval(1e-5)             # interpreter threads change hands every few bytecodes: the old race showed in most runs
    try:
        wrong = _run_fresh_pipelines(200, dg, train, fanouts, expected)
    finally:
        sys.setswitchinterval(interval)
    assert wrong == 0


def _run_fresh_pipelines(runs, dg, train, fanouts, expected):
    from dgll_amd.dataloader import DataLoader
    from dgll_amd.pipeline import MiniBatchPipeline
    from dgll_amd.sampling import FastNeighborSampler

    wrong = 0
    for run in range(runs):
        loader = DataLoader(dg, train, FastNeighborSampler(fanouts, defer_last_hop=bool(run & 1)), batch_size=64)
        pipe = MiniBatchPipeline(loader, labels=dg.labels, queue_size=3, device="cpu", hops="sampled", sampler_threads=4,
                                 base_seed=11, epoch=2)
        for b in pipe:
            inp, outp, layers = expected[b.step]
            ok = b.output_nodes.tolist() == outp and torch.as_tensor(b.input_nodes).tolist() == inp
            for sg, (src, dst) in zip(b.subgraphs, layers):
                ok = ok and sg.src_nodes().tolist() == src and sg.dst_nodes().tolist() == dst
            wrong += not ok
    return wrong


def test_sampler_csr_copy_is_safe_for_direct_callers_from_many_threads():
    """_csr without the pipeline's prepare(): 8 threads call sample_seeded on one fresh sampler at the same moment."""
    from dgll_amd.data import DGraph
    from dgll_amd.sampling import FastNeighborSampler
    from oracle import sampler as osampler

    n = 4000
    edges = _zipf_graph(n, 5)
    dg = DGraph(nodes=torch.arange(n), edges=edges, labels=torch.zeros(n, dtype=torch.long), features=torch.zeros(n, 1))
    seeds = torch.arange(0, 256)
    before = random.getstate()
    random.seed(99)
    want = osampler.sample(edges, seeds.tolist(), [4, 4])
    random.setstate(before)
    interval = sys.getswitchinterval()
    sys.setswitchinterval(1e-5)
    try:
        _hammer_one_sampler(dg, seeds, want)
    finally:
        sys.setswitchinterval(interval)


def _hammer_one_sampler(dg, seeds, want):
    from dgll_amd.sampling import FastNeighborSampler

    for _ in range(25):
        smp = FastNeighborSampler([4, 4])
        start = threading.Barrier(8)
        out = [None] * 8

        def work(t):
            start.wait()
            inp, outp, subgs = smp.sample_seeded(dg, seeds, 99)
            out[t] = (torch.as_tensor(inp).tolist(), [(sg.src_nodes().tolist(), sg.dst_nodes().tolist()) for sg in subgs])

        ts = [threading.Thread(target=work, args=(t,)) for t in range(8)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for got in out:
            assert got[0] == want[0] and got[1] == [(list(s), list(d)) for s, d in want[2]]


@pytest.mark.gpu
def test_threaded_pipeline_on_a_list_of_lists_dgraph_with_device_translation_is_bit_equal_per_batch(cuda_device):
    """The GPU twin of the regression above: the reference's own DGraph format (python list of lists), sampler_threads=4,
    device_graph= built from the SAME lists, cache + pinned rings + device-side translation of the outermost hop; 40 fresh
    samplers/pipelines; every batch: ids bit-equal to oracle/sampler.py, fetched rows == features[ids]."""
    from dgll_amd.cache import GraphCacheServer
    from dgll_amd.data import DGraph
    from dgll_amd.dataloader import DataLoader
    from dgll_amd.pipeline import MiniBatchPipeline
    from dgll_amd.sampling import FastNeighborSampler
    from dgll_amd.sampling.fast_sampler import batch_seed
    from oracle import sampler as osampler

    n = 1500
    edges = _zipf_graph(n, 3)
    feats = torch.randn(n, 8, generator=torch.Generator().manual_seed(1))
    dg = DGraph(nodes=torch.arange(n), edges=edges, labels=torch.arange(n) % 5, features=feats)
    deg = np.array([len(e) for e in edges], dtype=np.int64)
    indptr = torch.from_numpy(np.concatenate([[0], np.cumsum(deg)])).to(cuda_device)
    indices = torch.tensor([u for e in edges for u in e], dtype=torch.int64, device=cuda_device)
    fanouts = [5, 3]
    train = torch.randperm(n, generator=torch.Generator().manual_seed(0))[:512]
    before = random.getstate()
    expected = []
    for i in range(8):
        random.seed(batch_seed(11, 2, i))
        expected.append(osampler.sample(edges, train[i * 64:(i + 1) * 64].tolist(), fanouts))
    random.setstate(before)
    srv = GraphCacheServer(feats, n)
    srv.auto_cache(torch.from_numpy(deg), capacity=n // 2)
    interval = sys.getswitchinterval()
    sys.setswitchinterval(1e-5)
    try:
        for run in range(40):
            loader = DataLoader(dg, train, FastNeighborSampler(fanouts, defer_last_hop=True), batch_size=64)
            pipe = MiniBatchPipeline(loader, cache=srv, labels=dg.labels, queue_size=3, device=cuda_device, hops="sampled",
                                     sampler_threads=4, base_seed=11, epoch=2, device_graph=(indptr, indices), build_blocks=bool(run & 1))
            steps = []
            for b in pipe:
                inp, outp, layers = expected[b.step]
                torch.cuda.current_stream().synchronize()
                assert torch.as_tensor(b.output_nodes).cpu().tolist() == outp
                assert b.input_nodes.is_cuda and b.input_nodes.cpu().tolist() == inp
                assert torch.equal(b.features[-1].cpu().float(), feats[torch.tensor(inp, dtype=torch.int64)].reshape(len(inp), -1))
                assert torch.as_tensor(b.subgraphs[1].src_nodes()).cpu().tolist() == layers[1][0]
                steps.append(b.step)
            assert steps == list(range(8))
    finally:
        sys.setswitchinterval(interval)


class _FailingSampler:
    """FastNeighborSampler whose sample_seeded raises on one batch (by seed)."""

    def __init__(self, inner, bad_seed):
        self._inner, self._bad = inner, bad_seed
        self.fanouts = inner.fanouts
        self.defer_last_hop = inner.defer_last_hop

    def prepare(self, g):
        self._inner.prepare(g)

    def sample_seeded(self, g, seeds, seed, **kw):
        if seed == self._bad:
            raise RuntimeError("sampler failed on purpose")
        return self._inner.sample_seeded(g, seeds, seed, **kw)


@pytest.mark.parametrize("threads", [1, 4])
def test_a_failing_sampler_worker_surfaces_its_exception_instead_of_hanging_the_epoch(threads):
    """after a worker failed, the surviving workers kept sampling batches nobody consumed and the joins never
    returned.  A long epoch (200 batches >> queue + hand-over capacity), batch 7 raises: the consumer gets the batches before it,
    then the exception, within seconds; no pipeline thread stays behind."""
    from dgll_amd.data import DGraph
    from dgll_amd.dataloader import DataLoader
    from dgll_amd.pipeline import MiniBatchPipeline
    from dgll_amd.sampling import FastNeighborSampler
    from dgll_amd.sampling.fast_sampler import batch_seed

    n = 1500
    dg = DGraph(nodes=torch.arange(n), edges=_zipf_graph(n, 3), labels=torch.arange(n) % 5, features=torch.zeros(n, 2))
    train = torch.arange(n).repeat(3)[:200 * 16]
    loader = DataLoader(dg, train, _FailingSampler(FastNeighborSampler([3, 2]), batch_seed(0, 0, 7)), batch_size=16)
    pipe = MiniBatchPipeline(loader, labels=dg.labels, queue_size=2, device="cpu", hops="sampled", sampler_threads=threads)
    seen, result = [], {}

    def consume():
        try:
            for b in pipe:
                seen.append(b.step)
            result["error"] = None
        except RuntimeError as exc:
            result["error"] = str(exc)

    t = threading.Thread(target=consume, daemon=True)
    t.start()
    t.join(30)
    assert not t.is_alive(), "the pipeline hung after a sampler worker failed"
    assert result["error"] == "sampler failed on purpose"
    assert seen == list(range(len(seen))) and len(seen) <= 7
    time.sleep(0.2)
    assert not [th.name for th in threading.enumerate() if th.name.startswith("dgll-")]


def test_a_consumer_that_leaves_early_winds_the_producers_down():
    from dgll_amd.data import DGraph
    from dgll_amd.dataloader import DataLoader
    from dgll_amd.pipeline import MiniBatchPipeline
    from dgll_amd.sampling import FastNeighborSampler

    n = 1500
    dg = DGraph(nodes=torch.arange(n), edges=_zipf_graph(n, 3), labels=torch.arange(n) % 5, features=torch.zeros(n, 2))
    train = torch.arange(n).repeat(3)[:200 * 16]
    for threads in (0, 4):
        loader = DataLoader(dg, train, FastNeighborSampler([3, 2]), batch_size=16)
        pipe = MiniBatchPipeline(loader, labels=dg.labels, queue_size=2, device="cpu", hops="sampled", sampler_threads=threads)
        it = iter(pipe)
        assert next(it).step == 0 and next(it).step == 1
        it.close()
        time.sleep(0.2)
        assert not [th.name for th in threading.enumerate() if th.name.startswith("dgll-")]
        assert [b.step for b in pipe][:3] == [0, 1, 2]          # and the pipeline can be iterated again


def test_ordered_handoff_returns_batches_in_order_and_bounds_the_run_ahead():
    from dgll_amd.pipeline import OrderedHandoff

    h = OrderedHandoff(capacity=3)
    put_log = []

    def producer(t):
        for i in range(t, 20, 4):
            time.sleep(0.001 * ((i * 7) % 5))
            h.put(i, i * i)
            put_log.append(i)

    threads = [threading.Thread(target=producer, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    got = []
    for k in range(20):
        assert max(put_log + [0]) < k + 3 + 4          # nobody is more than capacity (+ one item in each producer's hand) ahead
        got.append(h.get())
    for t in threads:
        t.join()
    h.close(20)
    assert got == [i * i for i in range(20)] and h.get() is None


@pytest.mark.gpu
def test_consumed_batches_are_never_overwritten_under_a_slow_consumer(cuda_device):
    """A batch's tensors are allocated on the loading stream and read on the consumer's: with a consumer whose kernels lag far behind
    its host thread (a long spin kernel queued per batch) the loader must not write the next gathers into memory those queued kernels
    still read (the pipeline registers them with the caching allocator, record_stream).  Every batch's rows, read on the consumer's
    stream AFTER the lag, must still equal the host's."""
    from dgll_amd.cache import GraphCacheServer
    from dgll_amd.data import DGraph
    from dgll_amd.dataloader import DataLoader
    from dgll_amd.pipeline import MiniBatchPipeline
    from dgll_amd.sampling import FastNeighborSampler

    rng = torch.Generator().manual_seed(3)
    n, deg = 3000, 12
    idx = torch.randint(0, n, (n * deg,), generator=rng)
    ptr = torch.arange(0, n * deg + 1, deg)
    feats = torch.randn(n, 64, generator=rng)
    labels = torch.arange(n) % 5
    dg = DGraph.from_csr(ptr.numpy(), idx.numpy(), labels=labels, features=feats)
    srv = GraphCacheServer(feats, gpuid=0)
    srv.auto_cache(torch.full((n,), deg), capacity=n // 2)
    loader = DataLoader(dg, torch.randperm(n, generator=rng)[:2560], FastNeighborSampler([3, 3]), batch_size=64)
    pipe = MiniBatchPipeline(loader, cache=srv, labels=labels, queue_size=2, device=cuda_device)
    random.seed(1)
    compute = torch.cuda.Stream(cuda_device)
    checks = []
    with torch.cuda.stream(compute):
        for b in pipe:
            torch.cuda._sleep(2_000_000)                       # ~1 ms of queued work ahead of this batch's reads
            got = b.features[0].float().sum(1)                 # reads the loader's buffer on the consumer's stream, late
            checks.append((got, feats[b.input_nodes].sum(1)))
    torch.cuda.synchronize()
    assert len(checks) == 40
    for got, ref in checks:
        torch.testing.assert_close(got.cpu(), ref, rtol=1e-5, atol=1e-5)
