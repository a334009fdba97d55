"""Full-graph GraphSAGE (the bench path) on the GPU vs the oracle, and the partitioned engine at world_size 1."""
import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu


def _setup(dev, n_scale=10, fin=20, hidden=(32, 16)):
    from dgll_amd import nn as dnn
    from dgll_amd import synth

    torch.manual_seed(0)
    g_cpu = synth.rmat_graph(n_scale, 8, seed=2, device="cpu", symmetric=True, weighted=False, self_loops=False)
    model = dnn.GraphSage(fin, list(hidden), None)
    x = torch.randn(g_cpu.n_rows, fin)
    return g_cpu, model, x


def _oracle_forward(g_cpu, model, x):
    h = x.numpy()
    rp, col = g_cpu.rowptr.numpy(), g_cpu.col.numpy()
    for layer in model.gcn:
        agg = cref.spmm_csr(rp, col, None, h, reduce="mean")
        h = np.maximum(cref.gemm(h, layer.weight.detach().numpy()) + cref.gemm(agg, layer.neighborAgg.weight.detach().numpy()), 0)
    return h


def test_forward_graph_matches_oracle_and_autograd(cuda_device):
    g_cpu, model, x = _setup(cuda_device)
    ref = _oracle_forward(g_cpu, model, x)
    # CPU autograd of the same formula for the gradients
    xr = x.clone().requires_grad_()
    row = g_cpu.row_index()
    deg = g_cpu.degrees().clamp(min=1).float()
    h = xr
    for layer in model.gcn:
        agg = torch.zeros(g_cpu.n_rows, h.shape[1]).index_add_(0, row, h[g_cpu.col.long()]) / deg[:, None]
        h = torch.relu(h @ layer.weight + agg @ layer.neighborAgg.weight)
    gout = torch.randn_like(h)
    (h * gout).sum().backward()
    ref_grads = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()

    m = model.to(cuda_device)
    xd = x.to(cuda_device).requires_grad_()
    out = m.forward_graph(g_cpu.to(cuda_device), xd)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
    (out * gout.to(cuda_device)).sum().backward()
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=2e-3, atol=2e-4)
    for p, r in zip(m.parameters(), ref_grads):
        np.testing.assert_allclose(p.grad.cpu().numpy(), r.numpy(), rtol=2e-3, atol=5e-4)


def test_world1_partition_engine_equals_forward_graph(cuda_device):
    from dgll_amd import dist as ddist

    g_cpu, model, x = _setup(cuda_device)
    g = g_cpu.to(cuda_device)
    m = model.to(cuda_device)
    part = ddist.partition_contiguous(g, 1, 0)
    assert part.n_halo == 0 and part.n_interior == g.n_rows
    engine = ddist.DistGraph(part, cuda_device)
    xd = x.to(cuda_device)
    a = m.forward_graph(g, xd)
    b = engine.sage_forward(m, engine.permute_to_local(xd))
    assert torch.equal(a[part.order], b)
    # RaCoM at world 1 is the identity on the gradients
    b.sum().backward()
    before = [p.grad.clone() for p in m.parameters()]
    ddist.RaCoM(m.parameters(), cuda_device).all_reduce_and_wait()
    for p, q in zip(m.parameters(), before):
        assert torch.equal(p.grad, q)


def test_bf16_forward_graph_tracks_fp32(cuda_device):
    from dgll_amd import ops

    g_cpu, model, x = _setup(cuda_device, fin=100, hidden=(256, 47))
    g = g_cpu.to(cuda_device)
    m = model.to(cuda_device)
    xd = x.to(cuda_device)
    ref = m.forward_graph(g, xd)
    xb = ops.alloc_features(g.n_rows, 100, torch.bfloat16, cuda_device)
    xb.copy_(xd)
    out = m.forward_graph(g, xb)
    assert out.dtype == torch.bfloat16
    err = (out.float() - ref).abs().max() / ref.abs().max()
    assert float(err) < 3e-2, float(err)
    out.float().sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())


def test_split_k_weight_gradient(cuda_device):
    from dgll_amd import dense

    x = torch.randn(70_001, 24, device=cuda_device)
    g = torch.randn(70_001, 40, device=cuda_device)
    np.testing.assert_allclose(dense.grad_weight(x, g).cpu().numpy(), (x.double().t() @ g.double()).cpu().numpy(), rtol=1e-4, atol=1e-2)
