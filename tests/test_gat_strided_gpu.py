"""Second-generation GAT passes (csrc/gat_kernel.hpp) in the layouts the golden layers do not reach: per-head widths that are
not a power-of-two number of vectors, scores in the row padding (the in-row form of the single-head output layer), the
score-gradient epilogue -- against the C oracle (forward, gatconv.py:111-148) and against the same passes in the plain layout."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _graph(n, seed, device):
    from dgll_amd import synth

    g = synth.rmat_graph(int(np.ceil(np.log2(n))), 10, seed=seed, device="cpu", symmetric=True, weighted=False, self_loops=True)
    return g, g.to(device)


@pytest.mark.parametrize("heads,fo,dtype", [(1, 48, torch.bfloat16), (3, 24, torch.bfloat16), (8, 32, torch.bfloat16),
                                            (1, 12, torch.float32), (5, 8, torch.float32), (8, 8, torch.float32)])
def test_forward_any_head_width_matches_oracle(cuda_device, heads, fo, dtype):
    from dgll_amd import ops
    from oracle import cref

    g_cpu, g = _graph(1 << 10, heads * 100 + fo, cuda_device)
    n = g.n_rows
    torch.manual_seed(fo)
    h = (torch.randn(n, heads * fo) * 0.5).to(dtype)
    s, t = torch.randn(n, heads) * 0.5, torch.randn(n, heads) * 0.5
    want = cref.gat_fwd(g_cpu.rowptr.numpy(), g_cpu.col.numpy(), h.float().numpy(), s.numpy(), t.numpy(), heads, 0.2, apply_elu=True, mode=0)
    got = ops.gat_aggregate(g, h.to(cuda_device), s.to(cuda_device), t.to(cuda_device), heads, 0.2, apply_elu=True, mode=0)
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
    np.testing.assert_allclose(got.float().cpu().numpy(), want, rtol=tol, atol=tol)


def test_in_row_scores_equal_the_compact_layout(cuda_device):
    """One head, 47 -> 48 columns in 128-byte rows: with pack_scores the neighbour score t_j sits behind the row's last column
    and arrives with the gather (INROW kernels), {s_i, dd_i} behind the DN rows; without it they are compact arrays.  Same
    arithmetic: outputs and all gradients agree to the last bits that survive a different summation order of the weights."""
    from dgll_amd import ops

    _, g = _graph(1 << 11, 7, cuda_device)
    n = g.n_rows
    torch.manual_seed(1)
    base = torch.randn(n, 48, device=cuda_device) * 0.5
    A = torch.zeros(48, 2, device=cuda_device)
    A[:47, 0], A[:47, 1] = torch.randn(47, device=cuda_device) * 0.3, torch.randn(47, device=cuda_device) * 0.3
    gout = torch.randn(n, 48, device=cuda_device).to(torch.bfloat16)
    results = []
    for packed in (True, False):
        store = torch.zeros(n, 64, dtype=torch.bfloat16, device=cuda_device)     # rows on 128-byte lines, 32 bytes of padding
        h = store[:, :48]
        h.copy_(base)
        h.requires_grad_()
        Ap = A.clone().requires_grad_()
        out = ops.gat_layer(g, h, Ap, 1, 0.2, apply_elu=True, pack_scores=packed)
        gh, gA = torch.autograd.grad(out, (h, Ap), gout)
        results.append((out.float(), gh.float(), gA.float()))
    for a, b in zip(*results):
        assert torch.isfinite(a).all() and torch.isfinite(b).all()
        assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max()) + 1e-6


def test_gat_layer_node_equals_the_separate_nodes(cuda_device):
    """ops.gat_layer (scores + aggregation as one autograd node, the scores' gradient added in the transposed pass's epilogue)
    against skinny_linear + gat_aggregate (separate nodes: a [n, 16] x [16, 256] product and an add in the backward)."""
    from dgll_amd import dense, ops

    _, g = _graph(1 << 11, 3, cuda_device)
    n, heads, fo = g.n_rows, 8, 32
    torch.manual_seed(2)
    h0 = (torch.randn(n, heads * fo, device=cuda_device) * 0.5).to(torch.bfloat16)
    A0 = torch.zeros(heads * fo, 2 * heads, device=cuda_device)
    for k in range(heads):
        A0[k * fo:(k + 1) * fo, k] = torch.randn(fo, device=cuda_device) * 0.2
        A0[k * fo:(k + 1) * fo, heads + k] = torch.randn(fo, device=cuda_device) * 0.2
    gout = torch.randn(n, heads * fo, device=cuda_device).to(torch.bfloat16)
    h1, A1 = h0.clone().requires_grad_(), A0.clone().requires_grad_()
    out1 = ops.gat_layer(g, h1, A1, heads, 0.2, apply_elu=True)
    gh1, gA1 = torch.autograd.grad(out1, (h1, A1), gout)
    h2, A2 = h0.clone().requires_grad_(), A0.clone().requires_grad_()
    st = dense.skinny_linear(h2, A2)
    out2 = ops.gat_aggregate(g, h2, st[:, :heads], st[:, heads:], heads, 0.2, apply_elu=True, mode=0)
    gh2, gA2 = torch.autograd.grad(out2, (h2, A2), gout)
    assert float((out1.float() - out2.float()).abs().max()) <= 1e-2 * float(out2.float().abs().max())
    assert float((gh1.float() - gh2.float()).abs().max()) <= 3e-2 * float(gh2.float().abs().max())
    mask = A0 != 0                                     # only the block-diagonal entries are parameters
    assert float((gA1.float() - gA2.float())[mask].abs().max()) <= 3e-2 * float(gA2.float()[mask].abs().max())


@pytest.mark.parametrize("heads,fo,dtype", [(8, 32, torch.bfloat16), (3, 24, torch.bfloat16), (2, 16, torch.bfloat16), (4, 32, torch.float32),
                                            (8, 8, torch.float32), (2, 64, torch.float32), (1, 40, torch.float32)])
def test_row_score_passes_equal_the_gathered_score_passes(cuda_device, heads, fo, dtype, monkeypatch):
    """dgll_hip_gat_fwd_rowscore / dgll_hip_gat_bwd_rows_rowscore (t_j formed from the gathered row; 32-bit gather offsets; one
    exponential per lane and four edges where a head has four lanes; heads that leave lanes of their group idle) against the passes
    that gather T: same outputs and gradients, fp32 to rounding, bf16 to its storage rounding.  A rows-long row (long-row chunks)
    rides along."""
    from dgll_amd import ops, ops_edge
    from dgll_amd.graph import CSRGraph

    _, g0 = _graph(1 << 11, 7 + heads, cuda_device)
    n = g0.n_rows
    # one row of 700 edges: chunked by the plan (threshold 256), finished by the finalize kernel
    rp, col = g0.rowptr.cpu(), g0.col.cpu()
    extra = torch.randperm(n, generator=torch.Generator().manual_seed(1))[:700].to(col.dtype)
    deg0 = int(rp[1] - rp[0])
    col = torch.cat([extra, col[deg0:]])
    rp = torch.cat([rp[:1], rp[1:] + (700 - deg0)])
    g = CSRGraph(rp.to(cuda_device), col.to(cuda_device), None, n, n)
    torch.manual_seed(fo)
    h0 = (torch.randn(n, heads * fo, device=cuda_device) * 0.5).to(dtype)
    A0 = torch.zeros(heads * fo, 2 * heads, device=cuda_device)
    for k in range(heads):
        A0[k * fo:(k + 1) * fo, k] = torch.randn(fo, device=cuda_device) * 0.3
        A0[k * fo:(k + 1) * fo, heads + k] = torch.randn(fo, device=cuda_device) * 0.3
    if dtype == torch.bfloat16:
        A0 = A0.to(dtype).float()                    # both forms then see the same a2 (the row-score form rounds it to bf16)
    gout = torch.randn(n, heads * fo, device=cuda_device).to(dtype)

    def run(row_scores):
        monkeypatch.setattr(ops_edge, "ROW_SCORES", row_scores)
        monkeypatch.setattr(ops_edge, "ROW_SCORES_BWD", row_scores)
        h, A = h0.clone().requires_grad_(), A0.clone().requires_grad_()
        with ops.LaunchTimer() as timer:
            out = ops.gat_layer(g, h, A, heads, 0.2, apply_elu=True)
            gh, gA = torch.autograd.grad(out, (h, A), gout)
        tags = [k for k in timer.summary() if k[0] == "gat"]
        assert sum("rowscore" in k for k in tags) == (2 if row_scores else 0), tags
        return out.float(), gh.float(), gA.float()

    out_t, gh_t, gA_t = run(False)
    out_r, gh_r, gA_r = run(True)
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    mask = A0 != 0
    for name, a, b in (("out", out_r, out_t), ("grad_h", gh_r, gh_t), ("grad_A", gA_r[mask], gA_t[mask])):
        assert torch.isfinite(a).all(), name
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()), (name, float((a - b).abs().max()), float(b.abs().max()))


def test_row_score_entry_points_refuse_what_32_bit_offsets_cannot_address(cuda_device):
    """Both row-score entry points address the gathered rows by 32-bit byte offsets: more than 2^24 rows, or more than 4 GB, is
    DGLL_ERR_UNSUPPORTED with a message naming the form to take (checked before anything is launched: dummy pointers)."""
    from dgll_amd import _lib

    one = torch.zeros(64, device=cuda_device)
    p = one.data_ptr()
    # (DGLL_ERR_UNSUPPORTED = -3, include/dgll_hip.h)
    too_many_rows = _lib.lib.dgll_hip_gat_fwd_rowscore(None, None, p, p, p, 256, p, p, p, 256, 1, p, 10, (1 << 24) + 1, 8, 32, 0.2, 1, None, 0, 0, 0)
    assert too_many_rows == -3 and b"dgll_hip_gat_fwd_strided" in _lib.lib.dgll_hip_last_error()
    too_large = _lib.lib.dgll_hip_gat_bwd_rows_rowscore(None, None, p, p, p, 1024, p, p, p, 1024, p, 1024, 0, p, p, 1024, p, 16, p, 10,
                                                        (1 << 21), 8, 128, 0.2, 1, None, 0)        # 2 M rows x 4 KB = 8 GB
    assert too_large == -3 and b"dgll_hip_gat_bwd_rows_strided" in _lib.lib.dgll_hip_last_error()


def test_bf16_spgat_bench_configuration_against_the_storage_emulating_oracle(cuda_device):
    """The SpGAT bench configuration (8 heads x 32 -> 1 x 47, bf16, scores in the row padding, one autograd node per layer)
    against CPU autograd of the reference's formulas (gatconv.py:117-148, :194-199) with bf16 rounding applied where the GPU path
    stores a tensor: x, W and a as bf16 operands, h = x.W stored, each layer's output stored.  The stored activations must
    agree to one bf16 step almost everywhere.  Gradients: the aggregation path is within bf16 rounding (grad_h 2e-3 per pass,
    tools/gat_grad_precision.py); the score gradients are differences of nearly equal sums, ds_i = sum_j c_ij (DN_i.h_j + dd_i),
    formed from bf16-stored DN rows.  Round 3: dd_i = -DN_i.hp_i with hp_i recovered from the stored bf16 output row: 5e-3 per
    pass, 3.1-5.5e-3 (W) and 5.5e-3-1e-2 (a) here.  Round 4: dd_i from the pass's own dot products, sum_j w_ij (DN_i.h_j) / den_i
    (csrc/gat_kernel.hpp) -- the same stored h_j enter dd_i and every dot product, so the cancellation is exact to fp32 rounding:
    1.7e-3 per pass (grad_s, grad_t: the level of grad_h), and here 3.7-4.3e-3 (W), 4.2-6.3e-3 (a): what is left is the bf16
    rounding of the gradients handed from layer to layer.  Log-probabilities of a bf16 model are formed in fp32."""
    from dgll_amd import nn as dnn
    from dgll_amd import ops, synth

    dev = cuda_device
    torch.manual_seed(0)
    g_cpu = synth.rmat_graph(12, 12, seed=4, device="cpu", symmetric=True, weighted=False, self_loops=True)
    n, fin, nhid, heads, ncls, alpha = g_cpu.n_rows, 100, 32, 8, 47, 0.2
    model = dnn.SpGAT(fin, nhid, ncls, 0.0, alpha, heads).to(dev)
    x = torch.randn(n, fin)
    xb = ops.alloc_features(n, fin, torch.bfloat16, dev)
    xb.copy_(x.to(dev))
    gout = torch.randn(n, ncls)

    rnd = lambda t: t.to(torch.bfloat16).float()                     # noqa: E731
    class _Store(torch.autograd.Function):       # a tensor kept in bf16: value rounded forward, its gradient rounded backward
        @staticmethod
        def forward(ctx, t):
            return rnd(t)

        @staticmethod
        def backward(ctx, g):
            return rnd(g)

    store = _Store.apply

    row = torch.repeat_interleave(torch.arange(n), g_cpu.rowptr[1:] - g_cpu.rowptr[:-1])
    col = g_cpu.col.long()

    def layer(xin, W, a, elu):                                        # one head, the reference's formulas
        fo = W.shape[1]
        h = store(xin @ W)
        z = torch.nn.functional.leaky_relu((h @ a[0, :fo])[row] + (h @ a[0, fo:])[col], alpha)
        e = torch.exp(-z)
        den = torch.zeros(n).index_add_(0, row, e)
        hp = torch.zeros(n, fo).index_add_(0, row, e[:, None] * h[col]) / den[:, None]
        return torch.nn.functional.elu(hp) if elu else hp

    cx = xb.float().cpu().requires_grad_()
    ref_params = []
    outs = []
    for att in model.attentions:
        W, a = rnd(att.W.detach().cpu()).requires_grad_(), rnd(att.a.detach().cpu()).requires_grad_()
        ref_params.append((W, a))
        outs.append(layer(cx, W, a, True))
    hid = store(torch.cat(outs, dim=1))
    W, a = rnd(model.out_att.W.detach().cpu()).requires_grad_(), rnd(model.out_att.a.detach().cpu()).requires_grad_()
    ref_params.append((W, a))
    ref_out = torch.log_softmax(store(torch.nn.functional.elu(layer(hid, W, a, False))), dim=1)       # fp32 log-probabilities
    (ref_out * gout).sum().backward()

    xin = ops.alloc_features(n, fin, torch.bfloat16, dev)
    xin.copy_(xb)
    xin.requires_grad_()
    out = model(xin, g_cpu.to(dev))
    (out.float() * gout.to(dev)).sum().backward()

    got = out.detach().float().cpu()
    ref = ref_out.detach()
    err = (got - ref).abs()
    assert out.dtype == torch.float32                                 # log-probabilities of a bf16 model are formed in fp32
    tight = float((err <= 1e-4 * (1.0 + ref.abs())).float().mean())  # rows whose 47 stored activations all agree
    print("log-probabilities: %.4f within 1e-4, max |diff| %.3e" % (tight, float(err.max())))
    assert tight >= 0.9 and float(err.max()) <= 0.1

    def close(name, a_, ref, tol):
        a_, ref = a_.detach().float().cpu(), ref.detach()
        rel = float((a_ - ref).norm() / ref.norm())
        print("%-24s relative L2 error %.3e" % (name, rel))
        assert rel <= tol, (name, rel)

    for k, att in enumerate(list(model.attentions) + [model.out_att]):
        close("W of layer/head %d" % k, att.W.grad, ref_params[k][0].grad, 6e-3)
        close("a of layer/head %d" % k, att.a.grad, ref_params[k][1].grad, 8e-3)
    close("input features", xin.grad, cx.grad, 6e-3)


@pytest.mark.parametrize("concat", [True, False])
def test_gatconv_on_the_gpu_treats_an_edgeless_row_like_the_reference(cuda_device, concat):
    """gatConv masks with -9e15 and soft-maxes the whole row (gatconv.py:34-36): a node without any edge attends uniformly to ALL
    nodes.  The GPU layer (edges only, no dense N x N) must give that row the mean of Wh and route its gradient there; every other
    row and all parameter gradients as the dense formula.  sparseGatConv asserts on such a graph, as gatconv.py:141 does."""
    from dgll_amd.nn import gatConv, sparseGatConv

    torch.manual_seed(1)
    n = 40
    g = torch.Generator().manual_seed(0)
    adj = (torch.rand(n, n, generator=g) < 0.15).float()
    adj = ((adj + adj.T + torch.eye(n)) > 0).float()
    adj[7, :] = 0
    x = torch.randn(n, 9)
    layer = gatConv(9, 8, dropout=0.0, alpha=0.2, concat=concat).eval()

    def dense(xx, W, a):
        Wh = xx @ W
        e = torch.nn.functional.leaky_relu(Wh @ a[:8] + (Wh @ a[8:]).T, 0.2)
        hp = torch.softmax(torch.where(adj > 0, e, torch.full_like(e, -9e15)), dim=1) @ Wh
        return torch.nn.functional.elu(hp) if concat else hp

    xr = x.clone().requires_grad_()
    Wr, ar = layer.W.detach().clone().requires_grad_(), layer.a.detach().clone().requires_grad_()
    want = dense(xr, Wr, ar)
    gout = torch.randn(n, 8)
    (want * gout).sum().backward()

    layer = layer.to(cuda_device)
    xd = x.to(cuda_device).requires_grad_()
    got = layer(xd, adj.to(cuda_device))
    assert torch.isfinite(got).all()
    torch.testing.assert_close(got.cpu(), want.detach(), rtol=1e-4, atol=1e-4)
    (got * gout.to(cuda_device)).sum().backward()
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=2e-3, atol=2e-4)
    torch.testing.assert_close(layer.W.grad.cpu(), Wr.grad, rtol=2e-3, atol=2e-4)
    torch.testing.assert_close(layer.a.grad.cpu(), ar.grad, rtol=2e-3, atol=2e-4)

    sp = sparseGatConv(9, 8, dropout=0.0, alpha=0.2).eval().to(cuda_device)
    with pytest.raises(AssertionError):
        sp(x.to(cuda_device), adj.to(cuda_device))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_cross_entropy_of_the_activations_is_the_nll_of_the_models_log_softmax(cuda_device, dtype):
    """SpGAT.forward ends in log_softmax (gatconv.py:199) and the reference's loops apply F.nll_loss to it.  forward_activations()
    stops before the log_softmax; ops.cross_entropy on it is the same loss with the same parameter gradients (bench.py's gat step
    takes that route: one kernel per direction instead of log_softmax + gather and their backward passes)."""
    from dgll_amd import nn as dnn
    from dgll_amd import ops, synth

    dev = cuda_device
    g = synth.rmat_graph(11, 10, seed=2, device=dev, symmetric=True, weighted=False, self_loops=True)
    n = g.n_rows
    x = ops.alloc_features(n, 40, dtype, dev)
    x.copy_(torch.randn(n, 40, device=dev))
    labels = torch.randint(0, 47, (n,), device=dev)
    res = {}
    for fused in (False, True):
        torch.manual_seed(4)
        model = dnn.SpGAT(40, 16, 47, 0.0, 0.2, 4).to(dev)
        if fused:
            loss = ops.cross_entropy(model.forward_activations(x, g), labels, reduction="sum") * (1.0 / n)
        else:
            out = model(x, g)
            assert out.dtype == torch.float32
            loss = torch.nn.functional.nll_loss(out, labels, reduction="sum") * (1.0 / n)
        loss.backward()
        res[fused] = (float(loss), [p.grad.float().clone() for p in model.parameters()])
    tol = 1e-5 if dtype == torch.float32 else 2e-3
    assert res[True][0] == pytest.approx(res[False][0], rel=tol)
    for a, b in zip(res[True][1], res[False][1]):
        assert float((a - b).abs().max()) <= (1e-4 if dtype == torch.float32 else 2e-2) * float(b.abs().max()) + 1e-7
