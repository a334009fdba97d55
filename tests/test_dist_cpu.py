"""world_size-2 (and 3) gloo tests of the partitioned aggregation path on CPU: partition bookkeeping, halo exchange,
overlap structure, backward exchange and RaCoM bucket all-reduce.  The SpMM itself is injected (torch CPU sparse
product) because the product kernels are GPU-only; everything else is the product code."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _few_threads(world):
    """A worker process keeps to a few intra-op threads: on a 256-thread host eight ranks with the default thread count each spend
    their time in the thread pool (the GPU box: 91 s for a test that takes 5 s on 8 cores)."""
    torch.set_num_threads(max(1, min(4, (os.cpu_count() or 4) // max(int(world), 1))))


def _cpu_spmm(graph, x, val=None, reduce="sum"):
    v = val if val is not None else (graph.val if graph.val is not None else torch.ones(graph.nnz))
    v = v.to(x.dtype)
    if graph.n_rows == 0 or graph.n_cols == 0:
        return torch.zeros(graph.n_rows, x.shape[1], dtype=x.dtype)
    adj = torch.sparse_csr_tensor(graph.rowptr, graph.col.long(), v, (graph.n_rows, graph.n_cols))
    y = torch.sparse.mm(adj, x.contiguous())
    if reduce == "mean":
        y = y / graph.degrees().clamp(min=1).unsqueeze(1).to(y.dtype)
    return y


def _worker(rank, world, port, weighted, feat, form="p2p"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DGLL_EXCHANGE=form)
    _few_threads(world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dgll_amd import dist as ddist
        from dgll_amd import synth

        torch.manual_seed(0)
        full = synth.rmat_graph(8, 6, seed=5, device="cpu", symmetric=True, weighted=weighted)
        n = full.n_rows
        x = torch.randn(n, feat, dtype=torch.float64)
        gout = torch.randn(n, feat, dtype=torch.float64)
        if full.val is not None:
            full.val = full.val.double()
        # single-process reference on the full graph
        xr = x.clone().requires_grad_()
        ref = _cpu_spmm(full, xr, reduce="mean")
        (ref * gout).sum().backward()

        bounds = ddist.nnz_balanced_bounds(full, world) if rank >= 0 else None
        part = ddist.partition_contiguous(full, world, rank, bounds)
        assert part.local.n_rows == part.halo.n_rows == part.n_own and part.local.nnz + part.halo.nnz == part.nnz
        assert sum(part.recv_counts) == part.n_halo and part.recv_counts[rank] == 0
        engine = ddist.DistGraph(part, "cpu", spmm_fn=_cpu_spmm)
        assert engine.exchange.form == form
        engine.verify()                      # exchange lists + the start-up self-test through the chosen form, both directions
        assert engine.exchange.calls == 0    # the self-test leaves the traffic counters at zero
        blk = slice(part.own_begin, part.own_end)
        h = engine.permute_to_local(x[blk]).clone().requires_grad_()
        out = engine.aggregate(h, reduce="mean")
        np.testing.assert_allclose(out.detach().numpy(), ref.detach()[blk].numpy(), rtol=1e-5, atol=1e-6)  # 1/deg is fp32
        (out * gout[blk]).sum().backward()
        np.testing.assert_allclose(h.grad.numpy(), xr.grad[blk].numpy(), rtol=1e-5, atol=1e-6)
        # input features: halo rows placed once, then aggregated without communication
        placed = engine.place_input_halo(x[blk].clone())
        np.testing.assert_allclose(engine.aggregate_static(placed, "mean").numpy(), ref.detach()[blk].numpy(), rtol=1e-5, atol=1e-6)

        # RaCoM: one flattened bucket, averaged (MQGCN.py:63-64)
        w = torch.nn.Parameter(torch.ones(3, 2))
        b = torch.nn.Parameter(torch.ones(5))
        w.grad = torch.full((3, 2), float(rank + 1))
        b.grad = torch.full((5,), float(10 * (rank + 1)))
        r = ddist.RaCoM([w, b], "cpu")
        r.all_reduce_and_wait()
        mean = sum(range(1, world + 1)) / world
        assert torch.allclose(w.grad, torch.full((3, 2), mean)) and torch.allclose(b.grad, torch.full((5,), 10 * mean))
    finally:
        dist.destroy_process_group()


def _racom_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    _few_threads(world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dgll_amd import dist as ddist

        def run(staleness, sync_every, steps=6):
            torch.manual_seed(0)
            w = torch.nn.Parameter(torch.zeros(4))
            opt = ddist.RaCoMOptimizer(torch.optim.SGD([w], lr=1.0), [w], "cpu", staleness=staleness, sync_every=sync_every)
            trace = []
            for t in range(steps):
                w.grad = torch.full((4,), float((rank + 1) * (t + 1)))      # rank- and step-dependent gradient
                opt.step()
                trace.append(w.detach().clone())
            opt.flush()
            return trace, w.detach().clone()

        mean_rank = sum(range(1, world + 1)) / world
        # staleness 0 == DDP: after step t the weights moved by the mean gradient of every step so far
        trace, final = run(0, 0)
        for t, wt in enumerate(trace):
            assert torch.allclose(wt, torch.full((4,), -mean_rank * sum(range(1, t + 2))))
        # staleness 1: the update lags one step, flush() applies the rest; the final weights are identical
        trace1, final1 = run(1, 0)
        assert torch.allclose(trace1[0], torch.zeros(4))                       # nothing applied yet after step 1
        assert torch.allclose(trace1[2], torch.full((4,), -mean_rank * 3))      # steps 1 and 2 applied after step 3
        assert torch.allclose(final1, final)
        # periodic synchronisation drains the queue
        trace2, _ = run(1, 3)
        assert torch.allclose(trace2[2], torch.full((4,), -mean_rank * 6))      # step 3 is a sync step: all three applied
    finally:
        dist.destroy_process_group()


def test_racom_async_queue_and_periodic_sync():
    mp.spawn(_racom_worker, args=(2, _free_port()), nprocs=2, join=True)


@pytest.mark.parametrize("world,weighted,feat", [(2, False, 12), (2, True, 7), (3, False, 5), (8, False, 6)])
def test_partitioned_aggregation_matches_single_process(world, weighted, feat):
    mp.spawn(_worker, args=(world, _free_port(), weighted, feat), nprocs=world, join=True)


@pytest.mark.parametrize("world,feat", [(2, 9), (8, 6)])
def test_alltoall_exchange_form_gives_the_same_results(world, feat):
    """DGLL_EXCHANGE=alltoall: the same packed buffers through one all-to-all-v collective (the fallback for a grouped p2p form
    that misbehaves on first contact with RCCL): self-test, forward, transposed exchange and static placement all pass unchanged."""
    mp.spawn(_worker, args=(world, _free_port(), False, feat, "alltoall"), nprocs=world, join=True)


def _bad_lists_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    _few_threads(world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dgll_amd import dist as ddist
        from dgll_amd import synth

        full = synth.rmat_graph(7, 6, seed=2, device="cpu", symmetric=True, weighted=False)
        part = ddist.partition_contiguous(full, world, rank)
        engine = ddist.DistGraph(part, "cpu", spmm_fn=_cpu_spmm)
        # a transport that delivers the right NUMBER of rows but the wrong rows (here: every rank rotates what it sends)
        part.send_idx = torch.roll(part.send_idx, 1)
        with pytest.raises(RuntimeError, match="self-test"):
            engine.verify()
    finally:
        dist.destroy_process_group()


def _fallback_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DGLL_EXCHANGE="p2p")
    _few_threads(world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dgll_amd import dist as ddist
        from dgll_amd import synth

        full = synth.rmat_graph(7, 6, seed=2, device="cpu", symmetric=True, weighted=False)
        part = ddist.partition_contiguous(full, world, rank)
        engine = ddist.DistGraph(part, "cpu", spmm_fn=_cpu_spmm)
        if rank == 1:
            engine.exchange.fault_p2p = True           # ONE rank's point-to-point form misdelivers
        engine.verify()                                # ... every rank switches to the all-to-all form together and passes
        assert engine.exchange.form == "alltoall"
        torch.manual_seed(0)                            # the same features on every rank
        x = torch.randn(full.n_rows, 4, dtype=torch.float64)
        blk = slice(part.own_begin, part.own_end)
        out = engine.aggregate(x[blk].clone(), reduce="mean")
        np.testing.assert_allclose(out.numpy(), _cpu_spmm(full, x, reduce="mean")[blk].numpy(), rtol=1e-5, atol=1e-6)
    finally:
        dist.destroy_process_group()


def test_failed_p2p_self_test_moves_every_rank_to_the_alltoall_form():
    mp.spawn(_fallback_worker, args=(3, _free_port()), nprocs=3, join=True)


def test_start_up_self_test_catches_rows_that_arrive_in_the_wrong_place():
    mp.spawn(_bad_lists_worker, args=(2, _free_port()), nprocs=2, join=True)


def test_exchange_watchdog_reports_the_phase_of_an_exchange_that_never_completes():
    """The watchdog's own logic, without killing the test process: a pending item that never completes triggers on_timeout with
    its phase name; completed ones do not; DGLL_EXCHANGE rejects unknown forms."""
    import time

    from dgll_amd import dist as ddist

    wd = ddist.ExchangeWatchdog(timeout=0.2, rank=5)
    fired = []
    wd.on_timeout = lambda phase, age: fired.append((phase, age))
    with wd.guard("quick phase"):
        pass
    state = {"done": False}
    wd._ensure_thread()
    with wd._lock:
        wd._pending.append(("layer-3 halo exchange", lambda: state["done"], time.monotonic(), None))
    time.sleep(0.8)
    assert [f[0] for f in fired] == ["layer-3 halo exchange"] and fired[0][1] > 0.2
    # default mode "startup": the first `arm_first` exchanges are watched, a steady-state job is not (a process paused under a
    # debugger or a serialised counter run must not get a healthy job killed); "always" and "off" are the other two
    wd2 = ddist.ExchangeWatchdog(timeout=5.0, arm_first=3)
    assert wd2.mode == "startup"
    for k in range(5):
        assert wd2.armed() == (k < 3)
        with wd2.guard("exchange %d" % k):
            pass
    assert wd2.watched == 3
    with wd2.guard("self-test", force=True):             # the start-up self-test is watched whatever was counted before
        assert any(ph == "self-test" for ph, _p, _t, _e in wd2._pending)
    # a new exchange pattern (halo mode switched, DistGraph.resolve_halo_mode) is watched from its own first contact again
    assert not wd2.armed()
    wd2.rearm()
    assert wd2.armed() and wd2.watched == 0
    assert ddist.ExchangeWatchdog(timeout=5.0, mode="always", arm_first=0).armed()
    assert not ddist.ExchangeWatchdog(timeout=5.0, mode="off").armed()
    with pytest.raises(ValueError):
        ddist.ExchangeWatchdog(mode="sometimes")
    part = ddist.Partition()
    with pytest.raises(ValueError):
        ddist._Exchange(part, form="ring")


def test_partition_covers_every_edge_once():
    from dgll_amd import dist as ddist
    from dgll_amd import synth

    full = synth.rmat_graph(9, 8, seed=1, device="cpu", symmetric=True, weighted=False)
    world = 4
    parts = [ddist.partition_contiguous(full, world, r) for r in range(world)]
    assert sum(p.nnz for p in parts) == full.nnz
    assert sum(p.local.nnz + p.halo.nnz for p in parts) == full.nnz
    for r, p in enumerate(parts):
        for q, other in enumerate(parts):
            assert p.send_counts[q] == other.recv_counts[r]          # what I send to q is what q expects from me
        assert (int(p.local.col.max()) if p.local.nnz else -1) < p.n_own              # owned-column half
        assert (int(p.halo.col.max()) if p.halo.nnz else -1) < p.n_halo              # halo-column half
        assert p.inv_deg.shape == (p.n_own,)


def _rows_worker(rank, world, port):
    """partition_rows: every rank holds ONLY its own row block; the send lists come from one exchange of halo ids."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    _few_threads(world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dgll_amd import dist as ddist
        from dgll_amd import synth

        full = synth.products_like_graph("cpu", seed=2, n=900, n_undirected=7000, locality=0.7, n_blocks=6, exact=True)
        bounds = ddist.nnz_balanced_bounds(full, world)
        want = ddist.partition_contiguous(full, world, rank, bounds)          # the all-ranks-hold-everything construction
        e0, e1 = int(full.rowptr[bounds[rank]]), int(full.rowptr[bounds[rank + 1]])
        rowptr = full.rowptr[bounds[rank]:bounds[rank + 1] + 1].clone()       # NOT rebased: partition_rows does it
        col = full.col[e0:e1].clone()
        del full
        got = ddist.partition_rows(rowptr, col, None, bounds, rank)
        for name in ("n_own", "n_halo", "nnz", "recv_counts", "send_counts"):
            assert getattr(got, name) == getattr(want, name), name
        assert torch.equal(got.send_idx, want.send_idx)
        for a, b in ((got.local, want.local), (got.halo, want.halo), (got.send_reduce, want.send_reduce)):
            assert torch.equal(a.rowptr, b.rowptr) and torch.equal(a.col, b.col)
        assert torch.equal(got.inv_deg, want.inv_deg)
        engine = ddist.DistGraph(got, "cpu", spmm_fn=_cpu_spmm)
        engine.verify()                                                        # still passes (lists are consistent by construction)
        x = torch.randn(got.n_own, 6, dtype=torch.float64, generator=torch.Generator().manual_seed(rank))
        y_rows = engine.aggregate(x, reduce="mean")
        y_full = ddist.DistGraph(want, "cpu", spmm_fn=_cpu_spmm).aggregate(x, reduce="mean")
        assert torch.equal(y_rows, y_full)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_partition_from_own_row_block_equals_the_full_graph_construction(world):
    mp.spawn(_rows_worker, args=(world, _free_port()), nprocs=world, join=True)


def test_partition_rows_single_rank_needs_no_process_group():
    from dgll_amd import dist as ddist
    from dgll_amd import synth

    g = synth.rmat_graph(7, 5, seed=1, device="cpu", symmetric=True, weighted=False)
    p = ddist.partition_rows(g.rowptr, g.col, None, [0, g.n_rows], 0)
    assert p.n_halo == 0 and p.local.nnz == g.nnz and p.send_counts == [0]


def _ddp_worker(rank, world, port):
    """RaCoM with staleness 0 (sync period 1) must equal DistributedDataParallel bit for bit (SURVEY 8 f3;
    reference semantics avg = sum of grads / world_size, MQGCN.py:55-79)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    _few_threads(world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch.nn.parallel import DistributedDataParallel as DDP

        from dgll_amd import dist as ddist

        def make():
            torch.manual_seed(11)                                    # identical initial replicas
            return torch.nn.Sequential(torch.nn.Linear(8, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))

        gen = torch.Generator().manual_seed(100 + rank)              # every rank sees different data
        data = [(torch.randn(16, 8, generator=gen), torch.randint(0, 3, (16,), generator=gen)) for _ in range(5)]
        ref_model = make()
        ddp = DDP(ref_model)
        ref_opt = torch.optim.SGD(ddp.parameters(), lr=0.1, momentum=0.9)
        model = make()
        base = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9)
        # Two ranks: one addition per element, so the reduction order is the same whatever the transport does, and the
        # reference's "sum, then / world_size" equals DDP's "divide, then sum" exactly (division by 2 is exact): bit-equal.
        # More ranks: the all-reduce itself adds the ranks in an order that depends on where an element sits in the buffer
        # (ring chunks), and DDP lays its bucket out differently from RaCoM's flat parameter order -- the same reduction in a
        # different association: equal to rounding.
        pow2 = world == 2
        opt = ddist.RaCoMOptimizer(base, model.parameters(), "cpu", staleness=0, sync_every=1,
                                   average="reference" if world & (world - 1) == 0 else "ddp")
        for x, y in data:
            ref_opt.zero_grad()
            torch.nn.functional.cross_entropy(ddp(x), y).backward()
            ref_opt.step()
            opt.zero_grad()
            torch.nn.functional.cross_entropy(model(x), y).backward()
            opt.step()
            for a, b in zip(ref_model.parameters(), model.parameters()):
                if pow2:
                    assert torch.equal(a, b), "RaCoM(staleness=0) diverged from DDP"
                else:
                    torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7)
        # and the documented rule for the sync period (README.md:35: "based on graph size and GPU count")
        assert ddist.racom_sync_period(2_449_029, 8) == 2 and ddist.racom_sync_period(2_449_029, 2) == 8
        assert ddist.racom_sync_period(134_217_728, 8) == 14 and ddist.racom_sync_period(1000, 8) == 2
        assert ddist.racom_sync_period(10 ** 12, 1) == 64
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 3, 8])
def test_racom_sync_form_equals_ddp_bit_for_bit(world):
    mp.spawn(_ddp_worker, args=(world, _free_port()), nprocs=world, join=True)


def _sage_model_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    _few_threads(world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dgll_amd import dist as ddist
        from dgll_amd import nn as dnn
        from dgll_amd import synth
        from oracle import torch_ref

        torch.manual_seed(0)
        full = synth.rmat_graph(9, 6, seed=7, device="cpu", symmetric=True, weighted=False)
        n = full.n_rows
        model = dnn.GraphSage(12, [24, 24, 5], None).double()
        x = torch.randn(n, 12, dtype=torch.float64)
        gout = torch.randn(n, 5, dtype=torch.float64)
        # single-process reference: the oracle's layer formula on the whole graph
        h = x
        for layer in model.gcn:
            h = torch_ref.sage_block(full.rowptr, full.col, h, h, layer.weight, layer.neighborAgg.weight, act=layer.activation is not None)
        (h * gout).sum().backward()
        ref_out = h.detach()
        ref_grads = [p.grad.clone() for p in model.parameters()]

        part = ddist.partition_contiguous(full, world, rank, ddist.nnz_balanced_bounds(full, world))
        engine = ddist.DistGraph(part, "cpu", spmm_fn=_cpu_spmm)
        engine.verify()
        blk = slice(part.own_begin, part.own_end)
        results = {}
        for recompute in (True, False):
            model.zero_grad()
            engine.halo_recompute = recompute
            placed = engine.place_input_halo(x[blk].clone())
            out = engine.sage_forward(model, x[blk].clone(), placed)
            np.testing.assert_allclose(out.detach().numpy(), ref_out[blk].numpy(), rtol=5e-5, atol=1e-5)   # the engine's 1/deg is fp32
            (out * gout[blk]).sum().backward()
            ddist.RaCoM(model.parameters(), "cpu").all_reduce_and_wait()            # averages the ranks' (partial) gradients
            for p, r in zip(model.parameters(), ref_grads):
                np.testing.assert_allclose((p.grad * world).numpy(), r.numpy(), rtol=1e-4, atol=1e-5 * float(r.abs().max()))
            results[recompute] = [p.grad.clone() for p in model.parameters()]
        for a, b in zip(results[True], results[False]):
            np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-4, atol=1e-5 * float(b.abs().max()))   # host weight gradients are fp32
        # exchanges per step (forward + backward): the third layer's two with the recompute, the second layer's two more without
        calls = []
        start = engine.exchange.start
        engine.exchange.start = lambda *a, **k: (calls.append(tuple(a[0].shape)), start(*a, **k))[1]
        per_step = {}
        for recompute in (True, False):
            engine.halo_recompute = recompute
            placed = engine.place_input_halo(x[blk].clone())
            if recompute:
                engine.input_aggregate_all(placed, "mean")                            # the one-time placements, outside the step
            calls.clear()
            out = engine.sage_forward(model, x[blk].clone(), placed)
            (out * gout[blk]).sum().backward()
            per_step[recompute] = len(calls)
        assert per_step == {True: 2, False: 4}, per_step
        engine.exchange.start = start
        # DGLL_HALO_MODE=auto: both forms timed on the live ranks (max over ranks), every rank keeps the same, faster one
        os.environ["DGLL_HALO_MODE"] = "auto"
        try:
            eng2 = ddist.DistGraph(part, "cpu", spmm_fn=_cpu_spmm)
        finally:
            del os.environ["DGLL_HALO_MODE"]
        assert eng2.halo_mode == "auto" and eng2.halo_mode_timings is None
        placed2 = eng2.place_input_halo(x[blk].clone())

        def one_step():
            model.zero_grad()
            o = eng2.sage_forward(model, x[blk].clone(), placed2)
            (o * gout[blk]).sum().backward()

        chosen = eng2.resolve_halo_mode(one_step, reps=2)
        assert chosen in ("recompute", "exchange") and set(eng2.halo_mode_timings) == {"recompute", "exchange"}
        assert eng2.halo_recompute == (chosen == "recompute") and eng2.halo_mode == chosen
        votes = [None] * world
        dist.all_gather_object(votes, (chosen, eng2.halo_mode_timings))
        assert all(v == votes[0] for v in votes)                                     # one decision, the same numbers, on every rank
        assert (eng2.halo_mode_timings["recompute"] <= eng2.halo_mode_timings["exchange"]) == (chosen == "recompute")
        os.environ["DGLL_HALO_MODE"] = "sometimes"
        try:
            with pytest.raises(ValueError):
                ddist.DistGraph(part, "cpu", spmm_fn=_cpu_spmm)
        finally:
            del os.environ["DGLL_HALO_MODE"]
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_first_layer_recomputed_on_halo_rows_equals_the_exchange(world):
    """DistGraph.sage_forward with placed inputs: the first layer is evaluated on own + halo rows and the second aggregates over
    the merged adjacency, so layers 0 and 1 exchange nothing per step (dist._DistSageInputLayerAll).  Outputs and the all-reduced
    parameter gradients equal the single-process model and the exchanging path; the per-step exchanges left are the third layer's."""
    mp.spawn(_sage_model_worker, args=(world, _free_port()), nprocs=world, join=True)


# ---------------------------------------------------------------------------------------------------------------- config 5
def _rowblock_worker(rank, world, port, scale):
    """BASELINE config 5's sharding over gloo (process shape MQGCN.py:161-163): cost-balanced contiguous
    row blocks, X replicated, no exchange inside the aggregation; results == the single-process pass."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    _few_threads(world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dgll_amd import dist as ddist
        from dgll_amd import synth

        full = synth.rmat_graph(scale, 16, seed=3, device="cpu", symmetric=False, weighted=False, self_loops=True)
        full, _perm = full.reorder(method="degree", seed=0)            # hubs first, as bench.py orders RMAT-27
        n, feat = full.n_rows, 24
        x = torch.randn(n, feat, generator=torch.Generator().manual_seed(7))          # replicated: the same on every rank
        ref = _cpu_spmm(full, x, reduce="mean")

        def spmm(graph, xx, reduce):
            return _cpu_spmm(graph, xx, reduce=reduce)

        edge_cost, row_cost = feat * 4 + 4, feat * 4 + 8 + 2 * feat * 4
        shard = ddist.RowBlockShard(full, world, rank, edge_cost, row_cost, spmm_fn=spmm)
        # every rank derives the same cut points; blocks tile the rows; costs are balanced to within one (heaviest) row
        gathered = [None] * world
        dist.all_gather_object(gathered, (shard.bounds, shard.block.nnz, shard.n_own))
        assert all(b == shard.bounds for b, _, _ in gathered)
        assert sum(nz for _, nz, _ in gathered) == full.nnz and sum(r for _, _, r in gathered) == n
        assert shard.block_nnz == [nz for _, nz, _ in gathered]
        costs = [edge_cost * nz + row_cost * r for _, nz, r in gathered]
        heaviest_row = edge_cost * int(full.degrees().max()) + row_cost
        assert max(costs) - min(costs) <= 2 * heaviest_row, (costs, heaviest_row)
        equal_rows_nnz = [int(full.rowptr[(r + 1) * n // world] - full.rowptr[r * n // world]) for r in range(world)]
        assert max(equal_rows_nnz) > 1.5 * max(shard.block_nnz)          # what 2^scale / world blocks would have done on this order
        shard.own_copy()
        out = shard.aggregate(x, reduce="mean")
        assert out.shape == (shard.n_own, feat)
        assert torch.equal(out, ref[shard.own_begin:shard.own_end])      # the same rows, the same per-row order of additions
        everything = shard.gather_output(out)
        assert torch.equal(everything, ref)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,scale", [(2, 13), (8, 12)])
def test_config5_row_blocks_with_replicated_sources_equal_the_single_process_pass(world, scale):
    mp.spawn(_rowblock_worker, args=(world, _free_port(), scale), nprocs=world, join=True)


def test_cost_balanced_bounds_edge_cases():
    from dgll_amd import dist as ddist

    rp = torch.tensor([0, 0, 0, 10, 10, 11, 11, 11, 30], dtype=torch.int64)
    b = ddist.cost_balanced_bounds(rp, 3, edge_cost=1, row_cost=0)
    assert b[0] == 0 and b[-1] == 8 and b == sorted(b)
    assert ddist.cost_balanced_bounds(rp, 1) == [0, 8]
    # more parts than rows with edges: empty blocks are allowed, the cuts stay monotone
    b = ddist.cost_balanced_bounds(torch.tensor([0, 100], dtype=torch.int64), 4, 1, 0)
    assert b[0] == 0 and b[-1] == 1 and b == sorted(b)
    # rows only (no edges): equal row counts
    assert ddist.cost_balanced_bounds(torch.zeros(9, dtype=torch.int64), 4, 1, 1) == [0, 2, 4, 6, 8]
