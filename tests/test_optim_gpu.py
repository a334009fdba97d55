"""FlatAdam on the GPU: dgll_hip_adam_flat against torch.optim.Adam, the packed bf16 weight forms it emits against
dgll_hip_pack_weight_bf16, and the full-graph GraphSAGE step through it (weight gradients land in the slots, no pack launch)."""
import pytest
import torch

from dgll_amd import dense, nn as dnn, ops, synth
from dgll_amd.optim import FlatAdam

pytestmark = pytest.mark.gpu


def test_adam_kernel_matches_torch_adam(cuda_device):
    torch.manual_seed(0)
    shapes = [(100, 256), (256, 256), (256, 47), (47,), (3, 5)]
    a = [torch.nn.Parameter(torch.randn(*s, device=cuda_device)) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    ref = torch.optim.Adam(a, lr=3e-3, weight_decay=0.02)
    opt = FlatAdam(b, lr=3e-3, weight_decay=0.02)
    for step in range(6):
        grads = [torch.randn_like(p) * (0.1 + step) for p in a]
        for p, q, g in zip(a, b, grads):
            p.grad = g.clone()
            q.grad = g.clone()                      # not the slot: gather_grads copies it in
        ref.step()
        opt.step()
    for p, q in zip(a, b):
        torch.testing.assert_close(q, p, rtol=2e-6, atol=2e-7)
    # both packed forms follow the updates exactly (bf16 of the fp32 parameter, zero padding)
    for q in b:
        if q.dim() == 2 and max(q.shape) <= 256:
            assert torch.equal(dense._pad_wt(q.detach()), dense._pack_now(q.detach()))
            assert torch.equal(dense._pad_wt(q.detach().t()), dense._pack_now(q.detach().t()))
    # an in-place write by somebody else bumps the version: the lookup re-packs instead of serving a stale form
    with torch.no_grad():
        b[1].mul_(2.0)
    assert torch.equal(dense._pad_wt(b[1].t()), dense._pack_now(b[1].detach().t()))


def test_grad_scale_is_the_racom_average(cuda_device):
    torch.manual_seed(1)
    a = [torch.nn.Parameter(torch.randn(64, 32, device=cuda_device))]
    b = [torch.nn.Parameter(a[0].detach().clone())]
    ref, opt = torch.optim.Adam(a, lr=1e-2), FlatAdam(b, lr=1e-2)
    g = torch.randn(64, 32, device=cuda_device)
    a[0].grad = g / 8
    b[0].grad = g.clone()
    opt.grad_scale = 1.0 / 8
    ref.step()
    opt.step()
    torch.testing.assert_close(b[0], a[0], rtol=2e-6, atol=2e-7)


def test_full_graph_sage_step_through_flat_adam(cuda_device, monkeypatch):
    """The bench's step: same losses as with torch.optim.Adam, weight gradients written into the slots, no pack launch."""
    g = synth.products_like_graph(cuda_device, seed=3, n=20000, n_undirected=150000, locality=0.9, n_blocks=8, exact=True)
    x = ops.alloc_features(g.n_rows, 40, torch.bfloat16, cuda_device, pad_to=64)
    x.copy_(torch.randn(g.n_rows, 40, device=cuda_device))
    labels = torch.randint(0, 10, (g.n_rows,), device=cuda_device)
    losses = {}
    for kind in ("torch", "flat"):
        torch.manual_seed(5)
        model = dnn.GraphSage(40, [256, 256, 10], None).to(cuda_device)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2) if kind == "torch" else FlatAdam(model.parameters(), lr=1e-2)
        packs = []
        real = dense._pack_now
        monkeypatch.setattr(dense, "_pack_now", lambda wt, rows=None: (packs.append(1), real(wt, rows))[1])
        trace = []
        for it in range(4):
            opt.zero_grad(set_to_none=True)
            loss = ops.cross_entropy(model.forward_graph(g, x), labels, reduction="mean")
            loss.backward()
            if kind == "flat" and it == 0:
                slots = {id(p): opt.grad_slot(p).data_ptr() for p in opt.params}
                assert all(p.grad is not None and p.grad.data_ptr() == slots[id(p)] for p in opt.params)
            opt.step()
            trace.append(float(loss))
        monkeypatch.setattr(dense, "_pack_now", real)
        losses[kind] = trace
        if kind == "flat":
            assert not packs, "a FlatAdam-owned weight was packed by a separate launch"
        else:
            assert packs
    assert losses["flat"] == pytest.approx(losses["torch"], rel=2e-3)


def test_skipped_parameters_partial_launches_and_no_capture(cuda_device):
    """ADVICE round 4 on the GPU: a parameter without gradient is skipped like torch.optim.Adam skips it (value, moments, its step
    count); the others are updated by launches over their runs of the flat buffer, the packed bf16 forms of exactly those follow;
    step() refuses to be captured into a HIP graph."""
    torch.manual_seed(2)
    shapes = [(100, 256), (256, 256), (256, 47), (47,)]
    a = [torch.nn.Parameter(torch.randn(*s, device=cuda_device)) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    ref = torch.optim.Adam(a, lr=3e-3, weight_decay=0.02)
    opt = FlatAdam(b, lr=3e-3, weight_decay=0.02)
    for step in range(6):
        for i, (p, q) in enumerate(zip(a, b)):
            g = torch.randn_like(p)
            if i == 1 and step in (1, 2, 4):
                p.grad = q.grad = None
            else:
                p.grad, q.grad = g.clone(), g.clone()
        ref.step()
        opt.step()
    for p, q in zip(a, b):
        torch.testing.assert_close(q, p, rtol=2e-6, atol=2e-7)
    assert opt.param_steps == [6, 3, 6, 6]
    for q in b[:3]:
        assert torch.equal(dense._pad_wt(q.detach().t()), dense._pack_now(q.detach().t()))
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    for q in b:
        q.grad = torch.zeros_like(q)
    with pytest.raises(RuntimeError, match="cannot be captured"):
        with torch.cuda.graph(g, stream=side):
            opt.step()
    torch.cuda.synchronize()
