"""f1: DGraph / DGLLNeighborSampler / DataLoader reproduce the reference's IDs bit-for-bit (goldens captured by running
the reference under random.seed(s)); block construction; mini-batch GraphSage on the GPU."""
import random

import numpy as np
import pytest
import torch

from conftest import load_golden
from dgll_amd.data import DGraph
from dgll_amd.dataloader import DataLoader
from dgll_amd.sampling import DGLLNeighborSampler


def golden_graph():
    g = load_golden("sampler_n400")
    ptr, idx = g["adj_ptr"], g["adj_idx"]
    n = len(ptr) - 1
    edges = [idx[ptr[i]:ptr[i + 1]].tolist() for i in range(n)]
    dg = DGraph(nodes=torch.arange(n), edges=edges, labels=torch.arange(n) % 7,
                features=torch.arange(n * 3, dtype=torch.float32).view(n, 3))
    return g, dg


def test_sampled_ids_are_bit_exact():
    g, dg = golden_graph()
    for ci, case in enumerate(g.meta["cases"]):
        sampler = DGLLNeighborSampler(case["fanouts"])
        random.seed(case["seed"])
        if not case["fanouts"]:
            with pytest.raises(UnboundLocalError):
                sampler.sample(dg, torch.tensor(case["seeds"]))
            continue
        inp, outp, subgs = sampler.sample(dg, torch.tensor(case["seeds"]))
        assert torch.equal(inp, g.t("c%d_input_nodes" % ci))
        assert torch.equal(outp, g.t("c%d_output_nodes" % ci))
        assert len(subgs) == case["n_layers"]
        for li, sg in enumerate(subgs):
            assert torch.equal(sg.src_nodes(), g.t("c%d_l%d_src" % (ci, li)))
            assert torch.equal(sg.dst_nodes(), g.t("c%d_l%d_dst" % (ci, li)))
            assert torch.equal(sg.nodes(), g.t("c%d_l%d_nodes" % (ci, li)))
            # block bookkeeping: one row per seed occurrence, counts add up
            blk = sg.to_block("cpu")
            assert blk.nnz == sg.num_src_nodes() and int(blk.rowptr[-1]) == blk.nnz


def test_native_sampler_is_bit_exact_with_the_python_one():
    """dgll_host_sample_neighbors (C++ MT19937 + CPython's random.sample algorithm) vs the stdlib-driven sampler: same
    ids AND the same generator state afterwards, through both branches of random.sample (pool and set) and across
    several fan-outs; then against the reference goldens."""
    from dgll_amd.sampling import FastNeighborSampler

    rng = np.random.default_rng(0)
    n = 3000
    edges = []
    for v in range(n):
        deg = int(min(n - 1, rng.zipf(1.3))) if v % 11 else 0        # heavy tail: degrees up to n-1
        edges.append(rng.choice(n, size=deg, replace=False).tolist())
    dg = DGraph(nodes=torch.arange(n), edges=edges, labels=torch.zeros(n), features=torch.zeros(n, 1))
    assert max(len(e) for e in edges) > 1000
    for seed, fanouts in [(0, [3]), (1, [5, 4]), (2, [6, 25]), (3, [25, 10, 10]), (4, [100, 1])]:
        seeds = torch.from_numpy(rng.integers(0, n, 200))
        random.seed(seed)
        a = DGLLNeighborSampler(fanouts).sample(dg, seeds)
        state_a = random.getstate()
        random.seed(seed)
        b = FastNeighborSampler(fanouts).sample(dg, seeds)
        assert random.getstate() == state_a                          # the stream was consumed identically
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        for sa, sb in zip(a[2], b[2]):
            assert torch.equal(sa.src_nodes(), sb.src_nodes()) and torch.equal(sa.dst_nodes(), sb.dst_nodes())
            assert torch.equal(sa.indptr, sb.indptr)
    g, gdg = golden_graph()
    for ci, case in enumerate(g.meta["cases"]):
        if not case["fanouts"]:
            continue
        random.seed(case["seed"])
        inp, outp, subgs = FastNeighborSampler(case["fanouts"]).sample(gdg, torch.tensor(case["seeds"]))
        assert torch.equal(inp, g.t("c%d_input_nodes" % ci))
        for li, sg in enumerate(subgs):
            assert torch.equal(sg.src_nodes(), g.t("c%d_l%d_src" % (ci, li)))
            assert torch.equal(sg.dst_nodes(), g.t("c%d_l%d_dst" % (ci, li)))


def test_dgraph_queries_match_reference():
    g, dg = golden_graph()
    q = g.t("q_nodes")
    assert torch.equal(dg.get_induced_subgraph(q), g.t("q_induced"))
    assert torch.equal(dg.get_features(q), g.t("q_features"))
    assert torch.equal(dg.get_labels(q), g.t("q_labels"))
    assert dg.feature_size() == 3


def test_dataloader_batches_and_oversized_fanout():
    _, dg = golden_graph()
    train = torch.arange(0, 100)
    loader = DataLoader(dg, train, DGLLNeighborSampler([1000, 1000]), batch_size=32)
    random.seed(0)
    batches = list(loader)
    assert len(batches) == len(loader) == 4
    inp, outp, subgs = batches[-1]
    assert torch.equal(outp, train[96:])
    # fan-out larger than any degree: every neighbour is taken, no RNG draw (base_sampler.py:53-54)
    assert subgs[-1].num_src_nodes() == sum(len(dg.edges[v]) for v in outp.tolist())


@pytest.mark.gpu
def test_minibatch_graphsage_on_sampled_blocks(cuda_device):
    """graphage.py:47-59 loop shape: sample -> gather features -> GraphSage on the blocks -> loss -> backward; the
    forward is checked against the oracle's mean-SpMM + GEMM."""
    from dgll_amd import nn as dnn
    from oracle import cref

    _, dg = golden_graph()
    torch.manual_seed(0)
    dg.features = torch.randn(dg.num_nodes(), 24)
    model = dnn.GraphSage(24, [16, 8], [5, 5]).to(cuda_device)
    random.seed(3)
    inp, outp, subgs = DGLLNeighborSampler([5, 5]).sample(dg, torch.arange(40, 72))
    hops = [outp, subgs[1].src_nodes(), subgs[0].src_nodes()]           # hop 0, 1, 2 node lists
    feats = [dg.get_features(h).to(cuda_device) for h in hops]
    blocks = [subgs[1].to_block(cuda_device), subgs[0].to_block(cuda_device)]
    out = model.forward_sampled(feats, blocks)
    assert out.shape == (32, 8)
    # oracle
    hid = [f.cpu().numpy() for f in feats]
    ptrs = [subgs[1].indptr.numpy(), subgs[0].indptr.numpy()]
    for l, layer in enumerate(model.gcn):
        ws, wn = layer.weight.detach().cpu().numpy(), layer.neighborAgg.weight.detach().cpu().numpy()
        nxt = []
        for hop in range(2 - l):
            col = np.arange(hid[hop + 1].shape[0], dtype=np.int32)
            agg = cref.spmm_csr(ptrs[hop], col, None, hid[hop + 1], reduce="mean")
            nxt.append(np.maximum(cref.gemm(hid[hop], ws) + cref.gemm(agg, wn), 0))
        hid = nxt
    np.testing.assert_allclose(out.detach().cpu().numpy(), hid[0], rtol=1e-4, atol=1e-5)
    loss = torch.nn.functional.cross_entropy(out, dg.get_labels(outp).to(cuda_device))
    loss.backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())


def test_deferred_translation_gives_the_same_ids():
    """FastNeighborSampler(defer_last_hop=True) leaves the outermost hop as positions until first read: same ids, same
    generator state afterwards."""
    import random

    import numpy as np
    import torch

    from dgll_amd.data import DGraph
    from dgll_amd.sampling import FastNeighborSampler

    rng = np.random.default_rng(4)
    n = 500
    deg = rng.integers(0, 120, n)
    indptr = np.zeros(n + 1, np.int64)
    np.cumsum(deg, out=indptr[1:])
    indices = rng.integers(0, n, int(indptr[-1])).astype(np.int64)
    g = DGraph.from_csr(indptr, indices, labels=torch.zeros(n, dtype=torch.long), features=torch.zeros(n, 1))
    seeds = torch.from_numpy(rng.integers(0, n, 64))
    random.seed(11)
    a_in, a_out, a_sub = FastNeighborSampler([25, 10, 5]).sample(g, seeds)
    state_a = random.getstate()
    random.seed(11)
    b_in, b_out, b_sub = FastNeighborSampler([25, 10, 5], defer_last_hop=True).sample(g, seeds)
    assert random.getstate() == state_a                      # the generator never waits for the translation
    assert torch.equal(b_in.resolve(), a_in)
    for x, y in zip(a_sub, b_sub):
        assert torch.equal(x.src_nodes(), y.src_nodes()) and torch.equal(x.dst_nodes(), y.dst_nodes())
        assert torch.equal(x.indptr, y.indptr)


def test_per_batch_seeded_sampler_is_bit_exact_under_random_seed_and_thread_safe():
    """FastNeighborSampler.sample_seeded(g, seeds, s): the whole batch under its OWN generator, seeded as random.seed(s) seeds
    the interpreter's -- ids bit-equal to the reference loop (oracle/sampler.py = base_sampler.py:45-58 + dgllsampler.py:10-21)
    run right after random.seed(s), for small, huge and zero seeds (CPython's init_by_array over the 32-bit words of the int);
    the global generator is not touched; eight threads drawing different batches at once give the same ids as one after another."""
    import threading

    from dgll_amd.sampling import FastNeighborSampler
    from dgll_amd.sampling.fast_sampler import batch_seed
    from oracle import sampler as osampler

    rng = np.random.default_rng(1)
    n = 2500
    edges = []
    for v in range(n):
        deg = int(min(n - 1, rng.zipf(1.25))) if v % 13 else 0
        edges.append(rng.choice(n, size=deg, replace=False).tolist())
    dg = DGraph(nodes=torch.arange(n), edges=edges, labels=torch.zeros(n), features=torch.zeros(n, 1))
    fanouts = [7, 3, 25]

    def check(sampler, seeds, s, got):
        random.seed(s)
        inp, outp, layers = osampler.sample(edges, seeds.tolist(), fanouts)
        g_inp, g_outp, subgs = got
        g_inp = g_inp.resolve() if hasattr(g_inp, "resolve") else g_inp
        assert g_inp.tolist() == inp and g_outp.tolist() == outp
        for sg, (src, dst) in zip(subgs, layers):
            assert sg.src_nodes().tolist() == src and sg.dst_nodes().tolist() == dst
            assert int(sg.indptr[-1]) == len(src)

    for defer in (False, True):
        sampler = FastNeighborSampler(fanouts, defer_last_hop=defer)
        for s in (0, 1, 12345, 2 ** 32 - 1, 2 ** 32, batch_seed(7, 3, 11), 2 ** 70 + 5, -9):
            seeds = torch.from_numpy(rng.integers(0, n, size=64))
            random.seed(99)
            before = random.getstate()
            got = sampler.sample_seeded(dg, seeds, s)
            assert random.getstate() == before                     # the interpreter's generator is not involved
            check(sampler, seeds, s, got)
    # concurrency: every thread owns whole batches
    sampler = FastNeighborSampler(fanouts, defer_last_hop=True)
    batches = [torch.from_numpy(rng.integers(0, n, size=128)) for _ in range(16)]
    out = [None] * len(batches)

    def work(t):
        for b in range(t, len(batches), 8):
            out[b] = sampler.sample_seeded(dg, batches[b], batch_seed(0, 1, b))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for b, seeds in enumerate(batches):
        check(sampler, seeds, batch_seed(0, 1, b), out[b])
    assert batch_seed(7, 3, 11) == (7 << 40) | (3 << 20) | 11
    with pytest.raises(ValueError):
        batch_seed(0, 0, 1 << 20)


def test_staged_batch_holds_the_same_ids_in_one_buffer():
    """sample_seeded(staging=): the seeds, the source ids of every hop but the outermost and every hop's row pointers are written
    into ONE caller-provided buffer (the pipeline's pinned staging memory: one upload per batch) -- same ids as without it, found at
    the recorded offsets; staging_entries() is the size of the buffer for a full batch."""
    from dgll_amd.sampling import FastNeighborSampler
    from dgll_amd.sampling.fast_sampler import batch_seed

    rng = np.random.default_rng(4)
    n = 1500
    edges = [rng.choice(n, size=int(min(n - 1, rng.zipf(1.3))) if v % 11 else 0, replace=False).tolist() for v in range(n)]
    dg = DGraph(nodes=torch.arange(n), edges=edges, labels=torch.zeros(n), features=torch.zeros(n, 1))
    fanouts = [6, 4, 3]
    sampler = FastNeighborSampler(fanouts, defer_last_hop=True)
    handed = []

    def staging(entries):
        buf = np.full(entries + 7, -7, dtype=np.int64)
        handed.append((entries, buf))
        return buf[:entries], "token"

    for batch in (96, 17):
        seeds = torch.from_numpy(rng.integers(0, n, size=batch))
        s = batch_seed(1, 2, batch)
        plain = sampler.sample_seeded(dg, seeds, s)
        got = sampler.sample_seeded(dg, seeds, s, staging=staging)
        entries, buf = handed[-1]
        assert entries == FastNeighborSampler.staging_entries(batch, fanouts)
        assert (buf[entries:] == -7).all()                                    # nothing written past the requested entries
        L = len(fanouts)
        st = got[2][0].staged
        assert st is not None and st.token == "token" and all(sg.staged is st for sg in got[2])
        assert st.rows[0] == batch and len(st.rows) == L
        assert buf[:batch].tolist() == seeds.tolist()
        for h in range(L):
            sg_p, sg_s = plain[2][L - 1 - h], got[2][L - 1 - h]
            assert sg_s.indptr.tolist() == sg_p.indptr.tolist()
            o = st.offsets["ptr"][h]
            assert buf[o:o + st.rows[h] + 1].tolist() == sg_p.indptr.tolist()
            if h < L - 1:
                assert sg_s.src_nodes().tolist() == sg_p.src_nodes().tolist()
                o = st.offsets["src"][h]
                assert buf[o:o + len(sg_p.src_nodes())].tolist() == sg_p.src_nodes().tolist()
                assert st.rows[h + 1] == len(sg_p.src_nodes())
        assert got[0].resolve().tolist() == plain[0].resolve().tolist()


@pytest.mark.parametrize("threads", [1, 4])
def test_native_sampler_pool_delivers_every_batch_in_order_bit_equal_to_the_reference_loop(threads):
    """dgll_host_sampler_pool_* (pipeline.SamplerPool): K native threads draw the epoch's batches, each under
    random.seed(batch_seed(base, epoch, i)), into pinned slot buffers in the loading stage's upload layout; Python dequeues them in
    order.  Every batch -- seeds, the source ids and row pointers of every hop, the outermost hop's neighbours (handed over as
    positions, here translated on the host) -- equals the reference's loop (oracle/sampler.py = base_sampler.py:45-58 +
    dgllsampler.py:10-21) under that seed; more batches than slots (slots are reused as they are released), a ragged last batch."""
    from dgll_amd.pipeline import SamplerPool
    from dgll_amd.sampling.fast_sampler import batch_seed
    from oracle import sampler as osampler

    rng = np.random.default_rng(3)
    n = 3000
    edges = []
    for v in range(n):
        deg = int(min(n - 1, rng.zipf(1.3))) if v % 11 else 0
        edges.append(rng.choice(n, size=deg, replace=False).tolist())
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(e) for e in edges], out=indptr[1:])
    indices = np.array([u for e in edges for u in e], dtype=np.int64)
    fanouts = [6, 3, 9]                                   # the model's order; sampled in reversed order
    batch, n_batches = 50, 37
    train = torch.from_numpy(rng.integers(0, n, size=batch * n_batches - 17))
    max_deg = int(np.diff(indptr).max())
    pool = SamplerPool(indptr, indices, train, batch, fanouts, base_seed=5, epoch=2, n_threads=threads, max_degree=max_deg)
    assert pool.n_slots < n_batches and pool.pos_dtype == torch.int16
    L, off = pool.L, pool.offsets
    seen = 0
    try:
        while True:
            got = pool.next()
            if got is None:
                break
            i, slot, rows, n_src = got
            assert i == seen and slot == i % pool.n_slots
            seeds = train[i * batch:(i + 1) * batch]
            assert rows[0] == len(seeds)
            random.seed(batch_seed(5, 2, i))
            inp, outp, layers = osampler.sample(edges, seeds.tolist(), fanouts)      # layers: outermost first
            buf = pool.staged[slot]
            assert buf[off["seeds"]:off["seeds"] + rows[0]].tolist() == outp
            hop_seeds = outp
            for h in range(L):
                src, dst = layers[L - 1 - h]
                assert n_src[h] == len(src) and rows[h] == len(hop_seeds)
                ptr = buf[off["ptr"][h]:off["ptr"][h] + rows[h] + 1].numpy()
                assert ptr[0] == 0 and ptr[-1] == len(src)
                assert np.repeat(np.array(hop_seeds), np.diff(ptr)).tolist() == dst
                if h < L - 1:
                    got_src = buf[off["src"][h]:off["src"][h] + n_src[h]].tolist()
                else:                # positions inside every seed's adjacency list -> ids
                    pos = pool.pos[slot][:n_src[h]].numpy().astype(np.int64)
                    base = np.repeat(indptr[np.array(hop_seeds, dtype=np.int64)], np.diff(ptr))
                    got_src = indices[base + pos].tolist()
                assert got_src == src
                hop_seeds = src
            assert hop_seeds == inp
            seen += 1
            pool.release_part(slot, None)                 # both buffers: nothing to wait for on the host
            pool.release_part(slot, None)
        assert seen == n_batches and pool.next() is None and pool.delivered == n_batches
        pool.reap()
        assert pool.released == n_batches
    finally:
        pool.close()
    # a layout that cannot hold the upper bounds is refused, and so is a seed that does not fit the 64-bit word
    with pytest.raises(Exception):
        SamplerPool(indptr, indices, train, batch, fanouts, base_seed=1 << 24, epoch=0, n_threads=1, max_degree=max_deg)
