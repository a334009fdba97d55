"""The fused aggregate -> transform kernel (csrc/fused_sage.hip, dgll_hip_sage_fused_forward) against the C oracle's SpMM +
GEMM on bf16-rounded operands -- row a10 / a4 as one launch (gcn_fused_kernel.cu:5-74, sageconv.py:33-41,70-83): ragged
widths, a tile that ends mid-way, rows above the long-row threshold (chunk path -> agg_out), edge weights, the add-form of
the narrowing layer, and the layers that route through it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16).float()


def _reference(g_cpu, x, reduce, h, ws_t, wn_t, relu, bias=None):
    from oracle import cref

    agg = cref.spmm_csr(g_cpu.rowptr.numpy(), g_cpu.col.numpy(), None if g_cpu.val is None else g_cpu.val.numpy(),
                        _bf(x).numpy(), reduce=reduce)
    agg = _bf(torch.from_numpy(agg))                         # the kernel parks the aggregate as bf16 in LDS
    out = torch.zeros(g_cpu.n_rows, (wn_t if wn_t is not None else ws_t).shape[0])
    if h is not None:
        out = out + _bf(h) @ _bf(ws_t).T
    out = out + (agg @ _bf(wn_t).T if wn_t is not None else agg)
    if bias is not None:
        out = out + bias
    return (torch.relu(out) if relu else out), agg


@pytest.mark.parametrize("n,feat,k1,n_out,weighted,relu", [
    (1000, 256, 256, 256, False, True),      # the hidden layer of the benchmark
    (777, 100, 100, 256, False, True),       # first layer: ragged widths, last tile 9 rows
    (500, 47, 256, 47, False, False),        # narrowing layer: the aggregate is ADDED (no W2)
    (333, 64, 0, 128, True, True),           # relu(A.X.W) with edge weights, no self operand: the reference kernel's shape
    (64, 8, 24, 40, True, False),
])
def test_fused_kernel_matches_oracle(cuda_device, n, feat, k1, n_out, weighted, relu):
    from dgll_amd import dense, ops, synth

    torch.manual_seed(n)
    g_cpu = synth.rmat_graph(int(np.ceil(np.log2(n))), 12, seed=n, device="cpu", symmetric=True, weighted=weighted)
    g_cpu = type(g_cpu)(g_cpu.rowptr[:n + 1].clone(), (g_cpu.col[:int(g_cpu.rowptr[n])] % n).to(torch.int32),
                        None if g_cpu.val is None else g_cpu.val[:int(g_cpu.rowptr[n])].clone(), n, n, check=False)
    g = g_cpu.to(cuda_device)
    add_form = k1 > 0 and feat == n_out and feat == 47
    x = torch.randn(n, feat)
    h = torch.randn(n, k1) if k1 else None
    ws_t = torch.randn(n_out, k1) * 0.1 if k1 else None
    wn_t = None if add_form else torch.randn(n_out, feat) * 0.1
    bias = torch.randn(n_out) if not relu else None
    xd = ops.alloc_features(n, feat, torch.bfloat16, cuda_device)
    xd.copy_(x)
    hd = None
    if h is not None:
        hd = ops.alloc_features(n, k1, torch.bfloat16, cuda_device)
        hd.copy_(h)
    assert dense.fused_ok(g, xd, n_out, hd)
    out, agg = dense.sage_fused_forward(g, xd, "mean" if not weighted else "sum", hd,
                                        ws_t.to(cuda_device) if ws_t is not None else None,
                                        wn_t.to(cuda_device) if wn_t is not None else None, relu,
                                        bias=bias.to(cuda_device) if bias is not None else None, keep_agg=True)
    want, want_agg = _reference(g_cpu, x, "mean" if not weighted else "sum", h, ws_t, wn_t, relu, bias)
    torch.testing.assert_close(agg.float().cpu(), want_agg, rtol=2e-2, atol=2e-2)
    scale = float(want.abs().max()) + 1e-6
    assert float((out.float().cpu() - want).abs().max()) <= 2e-2 * scale


def test_fused_kernel_with_rows_above_the_long_row_threshold(cuda_device):
    """Hub rows (> 256 edges) are aggregated by the SpMM's chunk path into agg_out and loaded from there by the tile."""
    from dgll_amd import dense, ops
    from dgll_amd.graph import CSRGraph

    torch.manual_seed(5)
    n, feat = 600, 128
    deg = torch.randint(1, 30, (n,))
    deg[3], deg[130], deg[599] = 1500, 300, 257
    rowptr = torch.zeros(n + 1, dtype=torch.int64)
    torch.cumsum(deg, 0, out=rowptr[1:])
    col = torch.randint(0, n, (int(rowptr[-1]),), dtype=torch.int32)
    g_cpu = CSRGraph(rowptr, col, None, n, n)
    g = g_cpu.to(cuda_device)
    assert g.num_long_rows() == 3
    x, h = torch.randn(n, feat), torch.randn(n, feat)
    ws_t, wn_t = torch.randn(256, feat) * 0.1, torch.randn(256, feat) * 0.1
    xd, hd = x.to(cuda_device, torch.bfloat16), h.to(cuda_device, torch.bfloat16)
    out, agg = dense.sage_fused_forward(g, xd, "mean", hd, ws_t.to(cuda_device), wn_t.to(cuda_device), True, keep_agg=True)
    want, want_agg = _reference(g_cpu, x, "mean", h, ws_t, wn_t, True)
    torch.testing.assert_close(agg.float().cpu(), want_agg, rtol=2e-2, atol=2e-2)
    assert float((out.float().cpu() - want).abs().max()) <= 2e-2 * float(want.abs().max())
    # bit-reproducible (no atomics anywhere)
    out2, _ = dense.sage_fused_forward(g, xd, "mean", hd, ws_t.to(cuda_device), wn_t.to(cuda_device), True, keep_agg=True)
    assert torch.equal(out, out2)


def test_graphsage_forward_graph_is_the_same_with_and_without_the_fused_kernel(cuda_device):
    """GraphSage.forward_graph (sageconv.py:103-114 on the whole adjacency) through the fused launch and through SpMM + MFMA
    transform: activations and every gradient agree to bf16 rounding."""
    from dgll_amd import fused_layers, nn as dnn, ops, synth

    torch.manual_seed(0)
    g = synth.rmat_graph(12, 10, seed=2, device=cuda_device, symmetric=True, weighted=False)
    n = g.n_rows
    model = dnn.GraphSage(100, [256, 256, 47], None).to(cuda_device)
    x = ops.alloc_features(n, 100, torch.bfloat16, cuda_device, pad_to=64)
    x.copy_(torch.randn(n, 100, device=cuda_device))
    labels = torch.randint(0, 47, (n,), device=cuda_device)
    results = []
    for fuse in (True, False):
        fused_layers.FUSE_AGGREGATE_TRANSFORM = fuse
        try:
            model.zero_grad()
            out = model.forward_graph(g, x)
            ops.cross_entropy(out, labels).backward()
            results.append((out.float(), [p.grad.float().clone() for p in model.parameters()]))
        finally:
            fused_layers.FUSE_AGGREGATE_TRANSFORM = False
    (o1, g1), (o2, g2) = results
    assert float((o1 - o2).abs().max()) <= 3e-2 * float(o2.abs().max())
    for a, b in zip(g1, g2):
        assert float((a - b).abs().max()) <= 4e-2 * float(b.abs().max()) + 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sampled_forward_with_the_hops_stacked_equals_the_per_hop_loop(cuda_device, dtype):
    """GraphSage.forward_sampled stacks the hops of a layer (one transform, one pair of weight gradients per layer) when the blocks
    are aggregate-first; with batch_hops = False it runs the reference's per-(layer, hop) loop (sageconv.py:103-114).  Same outputs,
    same parameter and input gradients -- with separately allocated hop features, with hop features that are consecutive slices of
    one fetched buffer, and with the outermost hop arriving already reduced."""
    from dgll_amd import nn as dnn
    from dgll_amd import ops
    from dgll_amd.graph import CSRGraph

    dev = cuda_device
    torch.manual_seed(0)
    fan = [5, 3, 4]
    sizes = [64]
    for k in fan:
        sizes.append(sizes[-1] * k)
    F_in = 40
    model = dnn.GraphSage(F_in, [32, 32, 7], fan).to(dev)
    blocks = [CSRGraph.fixed_fanout(sizes[h], fan[h], dev) for h in range(3)]
    for b in blocks:
        b.identity_cols = True
    one = ops.alloc_features(sum(sizes[:3]), F_in, dtype, dev)
    one.copy_(torch.randn(sum(sizes[:3]), F_in, device=dev))
    outer = ops.alloc_features(sizes[3], F_in, dtype, dev)
    outer.copy_(torch.randn(sizes[3], F_in, device=dev))
    o = [0, sizes[0], sizes[0] + sizes[1], sum(sizes[:3])]
    views = [one[o[h]:o[h + 1]] for h in range(3)]
    tol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)

    def run(batch, feats, reduced=None):
        model.batch_hops = batch
        model.zero_grad()
        xs = [f.detach().clone().requires_grad_() if own else f for f, own in feats]
        out = model.forward_sampled([x for x in xs], blocks, last_hop_reduced=reduced)
        (out.float() ** 2).sum().backward()
        return out.detach().float(), [p.grad.clone() for p in model.parameters()], [x.grad for x in xs if x is not None and x.requires_grad]

    sep = [(v, True) for v in views] + [(outer, True)]
    ref_out, ref_gp, ref_gx = run(False, sep)
    out, gp, gx = run(True, sep)
    torch.testing.assert_close(out, ref_out, **tol)
    for a, b in zip(gp, ref_gp):
        torch.testing.assert_close(a, b, rtol=tol["rtol"], atol=tol["atol"] * float(b.abs().max()))
    for a, b in zip(gx, ref_gx):
        torch.testing.assert_close(a.float(), b.float(), rtol=tol["rtol"], atol=tol["atol"] * float(b.float().abs().max()))
    # consecutive slices of one buffer (no copy) + the outermost hop already reduced
    reduced = model.gcn[0].neighborAgg.reduce_block(blocks[2], outer).detach()
    out2, gp2, _ = run(True, [(v, False) for v in views] + [(None, False)], reduced)
    torch.testing.assert_close(out2, ref_out, **tol)
    for a, b in zip(gp2, ref_gp):
        torch.testing.assert_close(a, b, rtol=tol["rtol"], atol=tol["atol"] * float(b.abs().max()))
    assert dnn.GraphSage._stack_rows(views).data_ptr() == one.data_ptr()          # adjacency detected: a view, not a copy
    model.batch_hops = True
