"""dgll_amd.graphs.GraphedSampledStep: a sampled GraphSAGE step (graphage.py:47-63's loop body on sageconv.py:103-114's hop pyramid)
captured once on padded static block shapes and replayed per batch -- against the launch-by-launch step on the same batches."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _batch(dev, rng, batch, order, feats, classes, fill):
    """A pipeline-shaped batch with variable fan-out: hop h + 1 holds one row per sampled edge of hop h (duplicates kept, as the
    reference's sampler hands them on); `fill` in (0, 1]: share of the seed slots used."""
    from dgll_amd.graph import CSRGraph
    from dgll_amd import ops

    n = [max(1, int(batch * fill))]
    blocks = []
    for f in order:
        deg = torch.randint(0, f + 1, (n[-1],), generator=rng)
        deg[0] = f
        rowptr = torch.zeros(n[-1] + 1, dtype=torch.int64)
        torch.cumsum(deg, 0, out=rowptr[1:])
        n.append(int(rowptr[-1]))
        g = CSRGraph(rowptr.to(dev), torch.arange(n[-1], dtype=torch.int32, device=dev), None, n[-2], n[-1], check=False)
        g.identity_cols, g.max_degree = True, f
        blocks.append(g)
    L = len(order)
    store = ops.alloc_features(sum(n[:L]), feats, torch.bfloat16, dev)
    store.copy_(torch.randn(sum(n[:L]), feats, generator=rng).to(dev))
    offs = [0]
    for r in n[:L]:
        offs.append(offs[-1] + r)
    features = [store[offs[h]:offs[h + 1]] for h in range(L)] + [None]
    # the outermost hop arrives reduced: one row per hop L-1 node
    reduced = ops.alloc_features(n[L - 1], feats, torch.bfloat16, dev)
    reduced.copy_(torch.randn(n[L - 1], feats, generator=rng).to(dev))
    labels = torch.randint(0, classes, (n[0],), generator=rng).to(dev)
    return types.SimpleNamespace(features=features, last_hop_reduced=reduced, blocks=blocks[:L - 1] + [None], labels=labels)


def _eager(model, opt, b, ops):
    opt.zero_grad(set_to_none=True)
    loss = ops.cross_entropy(model.forward_sampled(b.features, b.blocks, last_hop_reduced=b.last_hop_reduced), b.labels)
    loss.backward()
    return float(loss), [p.grad.detach().clone() for p in model.parameters()]


def test_replayed_step_equals_the_eager_step_batch_by_batch(cuda_device):
    from dgll_amd import nn as dnn, ops
    from dgll_amd.graphs import GraphedSampledStep
    from dgll_amd.optim import FlatAdam

    dev = cuda_device
    fanouts, batch, feats, classes = [5, 4, 3], 64, 40, 7
    order = list(reversed(fanouts))
    rng = torch.Generator().manual_seed(0)
    batches = [_batch(dev, rng, batch, order, feats, classes, fill) for fill in (1.0, 1.0, 0.4, 1.0)]
    torch.manual_seed(1)
    model = dnn.GraphSage(feats, [32, 32, classes], fanouts).to(dev)
    opt = FlatAdam(list(model.parameters()), lr=0.0)                  # lr 0: the parameters stay put, every batch is comparable
    step = GraphedSampledStep(model, opt, batch, fanouts, feats, classes, device=dev)
    ref = [_eager(model, opt, b, ops) for b in batches]
    got = []
    for b in batches + [batches[0]]:                                   # ... and the first batch again, after a smaller one
        loss = step(b)
        got.append((float(loss), [p.grad.detach().clone() for p in model.parameters()]))
    for (lr_, gr), (lg, gg) in zip(ref + [ref[0]], got):
        assert lg == pytest.approx(lr_, rel=1e-5)
        for a, b_ in zip(gg, gr):
            assert float((a - b_).abs().max()) <= 2e-3 * float(b_.abs().max()) + 1e-7
    # rows a batch does not fill contribute exact zeros: the same batch gives the same bits whatever was loaded before it
    assert got[0][0] == got[-1][0]
    for a, b_ in zip(got[0][1], got[-1][1]):
        assert torch.equal(a, b_)
    with pytest.raises(ValueError):
        step.load(_batch(dev, rng, 2 * batch, order, feats, classes, 1.0))
    # tighter bounds than batch x prod(fan-outs) (what a caller's batches reach, plus a margin): same results for the batches that
    # fit, a refusal for one that does not
    need = [max(int(b.features[h].shape[0]) for b in batches) for h in range(3)]
    tight = GraphedSampledStep(model, opt, batch, fanouts, feats, classes, device=dev, rows=[batch, need[1] + 3, need[2] + 5])
    assert tight.rows[1] < step.rows[1] and tight.rows[2] < step.rows[2]
    for b, (lr_, gr) in zip(batches, ref):
        assert float(tight(b)) == pytest.approx(lr_, rel=1e-5)
        for a, b_ in zip([p.grad for p in model.parameters()], gr):
            assert float((a - b_).abs().max()) <= 2e-3 * float(b_.abs().max()) + 1e-7
    small = GraphedSampledStep(model, opt, batch, fanouts, feats, classes, device=dev, rows=[batch, need[1], max(need[2] // 2, 1)])
    with pytest.raises(ValueError):
        small.load(batches[0])


def test_replayed_training_follows_the_eager_loop(cuda_device):
    """Several optimizer steps: the captured forward / backward + FlatAdam's launch against the eager loop from the same start."""
    from dgll_amd import nn as dnn, ops
    from dgll_amd.graphs import GraphedSampledStep
    from dgll_amd.optim import FlatAdam

    dev = cuda_device
    fanouts, batch, feats, classes = [4, 3], 128, 24, 5
    order = list(reversed(fanouts))
    rng = torch.Generator().manual_seed(3)
    batches = [_batch(dev, rng, batch, order, feats, classes, fill) for fill in (1.0, 0.7, 1.0, 0.9, 1.0, 1.0)]
    losses = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(2)
        model = dnn.GraphSage(feats, [64, classes], fanouts).to(dev)
        opt = FlatAdam(list(model.parameters()), lr=1e-2)
        step = GraphedSampledStep(model, opt, batch, fanouts, feats, classes, device=dev) if mode == "graph" else None
        trace = []
        for b in batches:
            if step is not None:
                trace.append(float(step(b)))
            else:
                loss, _ = _eager(model, opt, b, ops)
                opt.step()
                trace.append(loss)
        losses[mode] = trace
    assert losses["graph"] == pytest.approx(losses["eager"], rel=5e-3)
    assert losses["graph"][-1] < losses["graph"][0] or losses["eager"][-1] >= losses["eager"][0]


def test_replay_equals_eager_at_a_fan_out_above_128(cuda_device):
    """PaddedBlock rewrites its row pointers under the captured step, but a launch plan with a long-row chunk
    schedule (fan-out > 128: graph.CSRGraph.plan scanned the capture-time pointers) would have described the synthetic full block
    on every replay -- wrong aggregates, silently.  Padded blocks now take the host-only plan whatever the fan-out; here a hop with
    fan-out 300 (rows of up to 300 edges, most batches far from full) replayed against the eager step."""
    from dgll_amd import nn as dnn, ops
    from dgll_amd.graphs import GraphedSampledStep
    from dgll_amd.optim import FlatAdam

    dev = cuda_device
    fanouts, batch, feats, classes = [3, 300], 24, 40, 5           # the model's order: hop 0 -> 1 was sampled with 300
    order = list(reversed(fanouts))
    rng = torch.Generator().manual_seed(5)
    batches = [_batch(dev, rng, batch, order, feats, classes, fill) for fill in (1.0, 0.5, 1.0)]
    assert max(int(b.blocks[0].degrees().max()) for b in batches) == 300
    torch.manual_seed(1)
    model = dnn.GraphSage(feats, [32, classes], fanouts).to(dev)
    opt = FlatAdam(list(model.parameters()), lr=0.0)
    step = GraphedSampledStep(model, opt, batch, fanouts, feats, classes, device=dev)
    assert step.blocks[0].max_degree == 300 and step.blocks[0].num_long_rows() == 0          # host-only plan: no chunk schedule
    for b in batches:
        want, grads = _eager(model, opt, b, ops)
        got = float(step(b))
        assert got == pytest.approx(want, rel=1e-5)
        for a, b_ in zip([p.grad for p in model.parameters()], grads):
            assert float((a - b_).abs().max()) <= 2e-3 * float(b_.abs().max()) + 1e-7


@pytest.mark.parametrize("native,pool", [(True, True), (True, False), (False, True), (True, "pageable")])
def test_batches_written_in_place_by_the_loading_stage_replay_to_the_same_losses(cuda_device, native, pool, monkeypatch):
    """MiniBatchPipeline.use_static_sets: the loading stage writes a batch's hop features (one cache gather per hop), the outermost
    hop's reduction, the row pointers and the labels straight into one of the captured step's input sets; the consumer replays that
    set's graph without a copy.  Same batches (per-batch seeds), parameters frozen: every in-place replay gives the loss the copy
    path gives for that batch, sets are handed round, and the launch-by-launch form on a static set agrees too."""
    import numpy as np

    # pool: the sampler threads are the native pool (round 6, the default) / Python threads around the native draw (the fallback)
    monkeypatch.setenv("DGLL_NATIVE_SAMPLER_POOL", "1" if pool else "0")
    # "pageable": the pool's slots are ordinary host memory -- the native loading call must notice (it uploads pinned slots with a kernel
    # that reads host memory) and take hipMemcpyAsync instead; same batches either way
    monkeypatch.setenv("DGLL_SAMPLER_POOL_PINNED", "0" if pool == "pageable" else "1")
    pool = bool(pool)
    from dgll_amd import nn as dnn, ops, synth
    from dgll_amd.cache import GraphCacheServer
    from dgll_amd.data import DGraph
    from dgll_amd.dataloader import DataLoader
    from dgll_amd.graphs import GraphedSampledStep
    from dgll_amd.optim import FlatAdam
    from dgll_amd.pipeline import MiniBatchPipeline
    from dgll_amd.sampling import FastNeighborSampler

    dev = cuda_device
    nodes, feats, classes, batch, fanouts = 20000, 50, 7, 64, [5, 3, 3]
    g = synth.products_like_graph(dev, seed=1, n=nodes, n_undirected=nodes * 12, locality=0.0, exact=True)
    indptr, indices = g.rowptr.cpu().numpy(), g.col.cpu().numpy().astype(np.int64)
    x = torch.randn(nodes, feats, generator=torch.Generator().manual_seed(0)).to(torch.bfloat16)
    labels = torch.randint(0, classes, (nodes,), generator=torch.Generator().manual_seed(1))
    dg = DGraph.from_csr(indptr, indices, labels=labels, features=x)
    cache = GraphCacheServer(x, gpuid=dev.index or 0)
    cache.auto_cache(g.degrees().cpu(), capacity=nodes // 2)
    n_batches = 14
    train = torch.randperm(nodes, generator=torch.Generator().manual_seed(2))[:n_batches * batch - 20]     # ragged last batch
    torch.manual_seed(3)
    model = dnn.GraphSage(feats, [32, 32, classes], fanouts).to(dev)
    opt = FlatAdam(list(model.parameters()), lr=0.0)
    dgraph = (torch.from_numpy(indptr).to(dev), torch.from_numpy(indices).to(dev))

    def pipeline():
        loader = DataLoader(dg, train, FastNeighborSampler(fanouts, defer_last_hop=True), batch_size=batch)
        return MiniBatchPipeline(loader, cache=cache, labels=labels, queue_size=2, device=dev, hops="sampled", reduce_last_hop="mean",
                                 sampler_threads=2, base_seed=5, epoch=0, device_graph=dgraph, build_blocks=True)

    step = GraphedSampledStep(model, opt, batch, fanouts, feats, classes, device=dev, n_sets=5)
    want = []
    for b in pipeline():                                    # the copy path (set 0)
        assert b.static_set is None
        want.append(float(step(b)))
    with pytest.raises(ValueError):
        pipeline().use_static_sets(GraphedSampledStep(model, opt, batch, fanouts, feats, classes, device=dev, n_sets=3))   # too few for the queue
    pipe = pipeline()
    pipe._native_loader = native            # True: ONE native call per batch (dgll_hip_load_sampled_batch); False: the torch-op path
    pipe.use_static_sets(step)
    got, used = [], []
    for i, b in enumerate(pipe):
        assert (pipe._pool is not None) == pool
        assert b.static_set is not None and 1 <= b.static_set <= 4
        used.append(b.static_set)
        if i in (0, n_batches - 1):             # the loaded arrays themselves, against a plain lookup (first and the ragged last batch)
            torch.cuda.current_stream().wait_event(b.ready)
            seeds = train[i * batch:(i + 1) * batch]
            assert torch.equal(b.features[0].float().cpu(), x[seeds].float()) and torch.equal(b.labels.cpu(), labels[seeds])
            assert torch.equal(b.features[1].float().cpu(), x[b.subgraphs[2].src_nodes().cpu()].float())
            ptr = b.subgraphs[0].indptr.cpu()
            ids = b.input_nodes.cpu()
            mean = torch.stack([x[ids[ptr[r]:ptr[r + 1]]].float().mean(0) if ptr[r + 1] > ptr[r] else torch.zeros(feats) for r in range(64)])
            assert float((b.last_hop_reduced[:64].float().cpu() - mean).abs().max()) < 2e-2
            blk = step.sets[b.static_set].blocks[0]
            n0 = int(b.features[0].shape[0])
            assert torch.equal(blk.rowptr[:n0 + 1].cpu(), b.subgraphs[2].indptr.cpu()) and bool((blk.rowptr[n0:] == blk.rowptr[n0]).all())
        got.append(float(step(b)) if i % 3 else float(step.eager(b)))      # every third batch launch by launch on its static set
    assert got == pytest.approx(want, rel=2e-3) and len(got) == n_batches
    assert used[:8] == [1, 2, 3, 4, 1, 2, 3, 4]
    torch.cuda.synchronize()


@pytest.mark.parametrize("streams,cap,feats", [(1, 0, 50), (2, 0, 50), (1, 9, 50), (1, 0, 64)])
def test_staged_misses_give_the_zero_copy_reduction_bit_for_bit(cuda_device, streams, cap, feats, monkeypatch):
    """csrc/gather.hip: list_misses_kernel + stage_rows_kernel fetch a batch's DISTINCT uncached rows (all hops) into HBM, the hop gathers
    and the outermost hop's reduction then read them there.  Same rows in the same order: every batch's gathered and reduced rows equal
    the zero-copy form's bit for bit -- on one loading stream and two alternating ones, and with a staging buffer of 9 rows (nodes past
    it stay zero-copy reads); 64 features: rows of whole 16-byte vectors (the 16-byte-lane forms of all three kernels)."""
    import numpy as np

    from dgll_amd import nn as dnn, pipeline as pl, synth
    from dgll_amd.cache import GraphCacheServer
    from dgll_amd.data import DGraph
    from dgll_amd.dataloader import DataLoader
    from dgll_amd.graphs import GraphedSampledStep
    from dgll_amd.optim import FlatAdam
    from dgll_amd.sampling import FastNeighborSampler

    dev = cuda_device
    nodes, classes, batch, fanouts = 12000, 5, 64, [5, 3, 4]
    g = synth.products_like_graph(dev, seed=2, n=nodes, n_undirected=nodes * 10, locality=0.0, exact=True)
    indptr, indices = g.rowptr.cpu().numpy(), g.col.cpu().numpy().astype(np.int64)
    x = torch.randn(nodes, feats, generator=torch.Generator().manual_seed(0)).to(torch.bfloat16)
    labels = torch.randint(0, classes, (nodes,), generator=torch.Generator().manual_seed(1))
    dg = DGraph.from_csr(indptr, indices, labels=labels, features=x)
    cache = GraphCacheServer(x, gpuid=dev.index or 0)
    cache.auto_cache(g.degrees().cpu(), capacity=nodes // 3)           # most draws of a low-degree graph miss: thousands per batch
    cache.log = True
    train = torch.randperm(nodes, generator=torch.Generator().manual_seed(2))[:9 * batch - 11]
    torch.manual_seed(3)
    model = dnn.GraphSage(feats, [16, 16, classes], fanouts).to(dev)
    opt = FlatAdam(list(model.parameters()), lr=0.0)
    dgraph = (torch.from_numpy(indptr).to(dev), torch.from_numpy(indices).to(dev))
    step = GraphedSampledStep(model, opt, batch, fanouts, feats, classes, device=dev, n_sets=5)

    def run(stage):
        monkeypatch.setattr(pl, "STAGE_MISSES", stage)
        monkeypatch.setattr(pl, "STAGE_CAP", cap)
        monkeypatch.setenv("DGLL_LOADER_STREAMS", str(streams))
        loader = DataLoader(dg, train, FastNeighborSampler(fanouts, defer_last_hop=True), batch_size=batch)
        pipe = pl.MiniBatchPipeline(loader, cache=cache, labels=labels, queue_size=2, device=dev, hops="sampled", reduce_last_hop="mean",
                                    sampler_threads=2, base_seed=5, epoch=0, device_graph=dgraph, build_blocks=True)
        assert len(pipe._load_streams) == streams
        pipe.use_static_sets(step)
        out, losses = [], []
        for b in pipe:
            assert b.static_set is not None
            torch.cuda.current_stream().wait_event(b.ready)
            every = torch.cat([sg.src_nodes().to(dev) for sg in b.subgraphs] + [train[len(out) * batch:(len(out) + 1) * batch].to(dev)])
            out.append((b.last_hop_reduced.clone(), b.input_nodes.clone(), [f.clone() for f in b.features if f is not None], every))
            losses.append(float(step(b)))
        torch.cuda.synchronize()
        stages = getattr(pipe, "_miss_stages", {})
        assert (len(stages) == streams) == stage
        for sg in stages.values():
            assert sg["cap"] == (cap or min(nodes - nodes // 3, batch * (1 + 4 + 12 + 60))) and sg["serial"] >= 9 // streams
        return out, losses, stages, cache.get_miss_rate()

    plain, plain_losses, _, plain_rate = run(False)
    staged, staged_losses, stages, staged_rate = run(True)
    assert len(plain) == len(staged) == 9
    for (a, ids_a, feats_a, _), (b_, ids_b, feats_b, _) in zip(plain, staged):
        assert torch.equal(ids_a, ids_b) and torch.equal(a, b_)
        assert len(feats_a) == len(feats_b) == 3 and all(torch.equal(u, v) for u, v in zip(feats_a, feats_b))      # the hop gathers too
    assert staged_losses == plain_losses
    assert staged_rate == pytest.approx(plain_rate) and plain_rate > 0.05         # the miss log counts draws, staged or not
    # the last batch of a stream: its distinct uncached nodes, each listed once
    sg = next(iter(stages.values()))
    count = int(sg["count"][0])
    slot = cache.localid2cacheid
    last_ids = staged[-1][3] if streams == 1 else None          # every id of the batch: seeds, each hop's sources, the outermost hop
    if last_ids is not None:
        distinct = torch.unique(last_ids[slot[last_ids] < 0])
        assert count == int(distinct.numel())
        listed = sg["list"][:min(count, sg["cap"])]
        assert int(torch.unique(listed).numel()) == int(listed.numel()) and bool(torch.isin(listed, distinct).all())
        rows = sg["rows"][:int(listed.numel()), :feats]
        assert torch.equal(rows.cpu(), x[listed.cpu()])


def test_sampled_step_gradients_written_in_place_equal_the_merged_ones(cuda_device):
    """ops.row_slices hands its consumers rows of ONE gradient buffer (the block's expand launch adds / writes there, the
    transform's self-path product writes there): the sampled step's parameter gradients equal the ones of the same step with that
    plumbing off (every range's gradient returned as its own tensor and merged by copies / adds), bit for bit in bf16."""
    from dgll_amd import nn as dnn, ops
    from dgll_amd.optim import FlatAdam

    dev = cuda_device
    fanouts, batch, feats, classes = [5, 4, 3], 96, 40, 7
    order = list(reversed(fanouts))
    rng = torch.Generator().manual_seed(11)
    b = _batch(dev, rng, batch, order, feats, classes, 1.0)
    torch.manual_seed(4)
    model = dnn.GraphSage(feats, [64, 64, classes], fanouts).to(dev)
    opt = FlatAdam(list(model.parameters()), lr=0.0)
    loss_a, grads_a = _eager(model, opt, b, ops)
    real = ops.grad_dest
    calls = []
    ops.grad_dest = lambda *a, **k: (calls.append(1), None)[1]        # plumbing off: producers return their own tensors
    import dgll_amd.dense as dense_mod
    try:
        loss_b, grads_b = _eager(model, opt, b, ops)
    finally:
        ops.grad_dest = real
    assert loss_a == loss_b and len(calls) >= 3
    for ga, gb in zip(grads_a, grads_b):
        assert float((ga - gb).abs().max()) <= 1e-6 * max(float(gb.abs().max()), 1.0)


@pytest.mark.parametrize("optimizer", ["flat", "torch"])
def test_every_set_steps_on_its_own_gradients(cuda_device, optimizer):
    """with n_sets > 1 every set's graph is captured after a zero_grad(set_to_none), so p.grad ends up bound to the LAST
    capture's tensors; a replay of an earlier set fills ITS tensors.  The step binds p.grad to the replayed set's gradient tensors
    before the optimizer runs: training with lr > 0 through sets handed round equals the eager loop, for FlatAdam (whose slots are the
    same tensor in every set) and for torch.optim.Adam (whose gradients live in each graph's pool)."""
    from dgll_amd import nn as dnn, ops
    from dgll_amd.graphs import GraphedSampledStep
    from dgll_amd.optim import FlatAdam

    dev = cuda_device
    fanouts, batch, feats, classes = [4, 3], 48, 30, 6
    order = list(reversed(fanouts))
    rng = torch.Generator().manual_seed(9)
    batches = [_batch(dev, rng, batch, order, feats, classes, fill) for fill in (1.0, 0.8, 1.0, 0.6, 1.0, 0.9, 1.0)]

    def make():
        torch.manual_seed(4)
        model = dnn.GraphSage(feats, [32, classes], fanouts).to(dev)
        params = list(model.parameters())
        return model, (FlatAdam(params, lr=2e-2) if optimizer == "flat" else torch.optim.Adam(params, lr=2e-2))

    model, opt = make()
    want, grads_want = [], []
    for b in batches:
        loss, grads = _eager(model, opt, b, ops)
        opt.step()
        want.append(loss)
        grads_want.append([g.clone() for g in grads])
    ref_params = [p.detach().clone() for p in model.parameters()]

    model, opt = make()
    step = GraphedSampledStep(model, opt, batch, fanouts, feats, classes, device=dev, n_sets=3)
    got = []
    for i, b in enumerate(batches):
        k = i % 3                                        # hand the sets round: load into set k by hand, as the loading stage would
        st = step.sets[k]
        if k:
            n = [int(b.features[h].shape[0]) for h in range(len(order))]
            for h in range(len(order)):
                st.features[h][:n[h]].copy_(b.features[h])
            st.reduced[:n[-1]].copy_(b.last_hop_reduced)
            from dgll_amd.graphs import PaddedBlock

            for h in range(len(order) - 1):
                PaddedBlock.pad(st.blocks[h], b.blocks[h].rowptr, n[h + 1])
            st.labels.fill_(-100)
            st.labels[:n[0]].copy_(b.labels)
            b.static_set = k
        else:
            b.static_set = None
        got.append(float(step(b)))
        for p, gw in zip(model.parameters(), grads_want[i]):       # the optimizer stepped on THIS batch's gradients
            assert float((p.grad - gw).abs().max()) <= 5e-3 * float(gw.abs().max()) + 1e-6, (i, k)
    assert got == pytest.approx(want, rel=5e-3)
    for p, r in zip(model.parameters(), ref_params):
        assert float((p.detach() - r).abs().max()) <= 2e-2 * float(r.abs().max())
    # one pool for all sets
    assert all(s.graph.pool() == step.sets[0].graph.pool() for s in step.sets)
