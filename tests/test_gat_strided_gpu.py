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
