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
