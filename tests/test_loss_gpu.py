"""dgll_hip_softmax_xent (ops.cross_entropy) against torch's cross_entropy on fp32 copies of the same logits."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("classes", [2, 7, 47, 64, 121, 300, 1000])
@pytest.mark.parametrize("reduction", ["mean", "sum", "none"])
def test_cross_entropy_matches_torch(cuda_device, dtype, classes, reduction):
    from dgll_amd import ops

    torch.manual_seed(classes)
    n = 1037
    z = (torch.randn(n, classes, device=cuda_device) * 4).to(dtype)
    labels = torch.randint(0, classes, (n,), device=cuda_device)
    labels[::13] = -100                                              # ignored targets
    za = z.clone().requires_grad_()
    zr = z.float().clone().requires_grad_()
    loss = ops.cross_entropy(za, labels, reduction=reduction)
    ref = F.cross_entropy(zr, labels, reduction=reduction)
    torch.testing.assert_close(loss, ref, rtol=1e-5, atol=1e-5)
    w = torch.randn_like(ref)
    (loss * w).sum().backward()
    (ref * w).sum().backward()
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    torch.testing.assert_close(za.grad.float(), zr.grad, rtol=tol, atol=tol * float(zr.grad.abs().max()))
    assert bool((za.grad[::13] == 0).all())


@pytest.mark.parametrize("n", [1, 255, 65536, 65537])
def test_cross_entropy_mean_on_both_sides_of_the_one_launch_bound(cuda_device, n):
    """Up to ops.XENT_ONE_LAUNCH_ROWS rows the batch's {total, count, mean, 1 / count} come from ONE native launch
    (dgll_hip_xent_reduce); above, from the two-stage tree of tensor ops: same loss and gradient as torch on both sides, ignored labels
    counted out, and an all-ignored batch is NaN as in torch."""
    from dgll_amd import ops

    assert ops.XENT_ONE_LAUNCH_ROWS == 65536
    torch.manual_seed(n)
    z = torch.randn(n, 41, device=cuda_device) * 3
    labels = torch.randint(0, 41, (n,), device=cuda_device)
    if n > 4:
        labels[::5] = -100
    for reduction in ("mean", "sum"):
        za, zr = z.clone().requires_grad_(), z.clone().requires_grad_()
        loss = ops.cross_entropy(za, labels, reduction=reduction)
        ref = F.cross_entropy(zr, labels, reduction=reduction)
        torch.testing.assert_close(loss, ref, rtol=2e-5, atol=1e-5)
        loss.backward(); ref.backward()
        torch.testing.assert_close(za.grad, zr.grad, rtol=1e-5, atol=1e-6 * max(float(zr.grad.abs().max()), 1e-3))
    ignored = torch.full((n,), -100, device=cuda_device)
    assert torch.isnan(ops.cross_entropy(z, ignored)) and torch.isnan(F.cross_entropy(z, ignored))


def test_cross_entropy_padded_rows_and_errors(cuda_device):
    from dgll_amd import ops

    store = torch.full((100, 64), float("nan"), device=cuda_device, dtype=torch.bfloat16)
    z = store[:, :47]
    z.copy_(torch.randn(100, 47, device=cuda_device))
    labels = torch.randint(0, 47, (100,), device=cuda_device)
    torch.testing.assert_close(ops.cross_entropy(z, labels), F.cross_entropy(z.float(), labels), rtol=1e-5, atol=1e-5)
    with pytest.raises(ValueError):
        ops.cross_entropy(z, labels[:5])
    with pytest.raises(RuntimeError):
        ops.cross_entropy(torch.randn(4, 2000, device=cuda_device), torch.zeros(4, dtype=torch.long, device=cuda_device))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("classes,n", [(121, 3230), (5, 70000), (300, 513)])
@pytest.mark.parametrize("reduction", ["mean", "sum", "none"])
def test_soft_target_cross_entropy_matches_torch(cuda_device, dtype, classes, n, reduction):
    """nn.CrossEntropyLoss with a float [N, C] target (multi-hot PPI labels, train_gcn.py:27,45)."""
    from dgll_amd import ops

    torch.manual_seed(classes + n)
    z = (torch.randn(n, classes, device=cuda_device) * 3).to(dtype)
    t = (torch.rand(n, classes, device=cuda_device) < 0.3).float()
    t[7] = 0                                                          # a row without any label
    za = z.clone().requires_grad_()
    zr = z.float().clone().requires_grad_()
    loss = ops.cross_entropy(za, t, reduction=reduction)
    ref = F.cross_entropy(zr, t, reduction=reduction)
    torch.testing.assert_close(loss, ref, rtol=2e-5, atol=1e-4)
    w = torch.randn_like(ref)
    (loss * w).sum().backward()
    (ref * w).sum().backward()
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    torch.testing.assert_close(za.grad.float(), zr.grad, rtol=tol, atol=tol * float(zr.grad.abs().max()))


def test_loss_that_folds_the_last_relu_gives_the_same_gradients(cuda_device):
    """ops.cross_entropy(fold_relu=True) on the tagged output of a full-graph GraphSage (whose last sageConv ends in a ReLU, as the
    reference's does, sageconv.py:83): the loss's gradient pass applies that ReLU's mask, the layer skips its own masking pass
    (ops.GateToken) -- identical losses and parameter gradients; untagged logits ignore the flag; the ex entry point masks where the
    logit is <= 0 and nowhere else."""
    from dgll_amd import nn as dnn, ops, synth

    dev = cuda_device
    g = synth.products_like_graph(dev, seed=2, n=5000, n_undirected=40000, locality=0.9, n_blocks=4, exact=True)
    x = ops.alloc_features(g.n_rows, 24, torch.bfloat16, dev, pad_to=64)
    x.copy_(torch.randn(g.n_rows, 24, device=dev))
    labels = torch.randint(0, 11, (g.n_rows,), device=dev)
    grads = {}
    for fold in (False, True):
        torch.manual_seed(3)
        model = dnn.GraphSage(24, [64, 64, 11], None).to(dev)
        out = model.forward_graph(g, x)
        token = getattr(out, "_dgll_gate_token", None)
        assert token is not None and not token.masked
        loss = ops.cross_entropy(out, labels, reduction="sum", fold_relu=fold) * (1.0 / g.n_rows)
        loss.backward()
        assert token.masked is fold
        grads[fold] = (float(loss), [p.grad.clone() for p in model.parameters()])
    assert grads[True][0] == grads[False][0]
    for a, b in zip(grads[True][1], grads[False][1]):
        assert torch.equal(a, b)
    # untagged logits: the flag does nothing
    z = torch.relu(torch.randn(300, 11, device=dev)).requires_grad_()
    ops.cross_entropy(z, labels[:300], fold_relu=True).backward()
    ref = z.detach().clone().requires_grad_()
    torch.nn.functional.cross_entropy(ref, labels[:300]).backward()
    torch.testing.assert_close(z.grad, ref.grad, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("classes", [2, 7, 8, 47, 64, 121, 300, 512])
def test_cross_entropy_on_16_byte_aligned_bf16_rows(cuda_device, classes):
    """bf16 logits whose rows start on 16-byte boundaries (what the layers hand the loss: a [N, C] view on a padded pitch) take the
    kernel with eight consecutive classes per lane: loss and gradient as torch's on fp32 copies, NaN-filled row padding never read
    into the result, ignored labels, the ReLU-folding mask."""
    from dgll_amd import ops

    torch.manual_seed(classes)
    n = 2051
    pitch = -(-classes // 64) * 64
    store = torch.full((n, pitch), float("nan"), device=cuda_device, dtype=torch.bfloat16)
    z = store[:, :classes]
    z.copy_(torch.randn(n, classes, device=cuda_device) * 3)
    labels = torch.randint(0, classes, (n,), device=cuda_device)
    labels[::17] = -100
    za = z.detach().requires_grad_()
    zr = z.detach().float().clone().requires_grad_()
    loss = ops.cross_entropy(za, labels, reduction="sum")
    ref = F.cross_entropy(zr, labels, reduction="sum")
    torch.testing.assert_close(loss, ref, rtol=2e-5, atol=1e-4)
    loss.backward()
    ref.backward()
    torch.testing.assert_close(za.grad.float(), zr.grad, rtol=1e-2, atol=1e-2 * float(zr.grad.abs().max()))
    assert bool((za.grad[::17] == 0).all()) and bool(torch.isfinite(za.grad).all())
    # the gradient pass with the ReLU mask folded in (dgll_hip_softmax_xent_ex, flags bit 0) on the same kernel
    from dgll_amd.ops import _xent_launch, alloc_features

    zp = torch.relu(z.detach())
    store2 = torch.zeros((n, pitch), device=cuda_device, dtype=torch.bfloat16)
    zz = store2[:, :classes]
    zz.copy_(zp)
    g = alloc_features(n, classes, torch.bfloat16, cuda_device, pad_to=64)
    _xent_launch(zz, labels, False, None, g, None, mask_nonpositive=True)
    g0 = alloc_features(n, classes, torch.bfloat16, cuda_device, pad_to=64)
    _xent_launch(zz, labels, False, None, g0, None)
    assert torch.equal(g, torch.where(zz > 0, g0, torch.zeros_like(g0)))
