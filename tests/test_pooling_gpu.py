"""Global pooling against the restated torch_scatter semantics (oracle/torch_ref.scatter_pool)."""
import pytest
import torch

from oracle import torch_ref

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("sorted_batch", [True, False])
def test_pooling_matches_scatter_semantics(dtype, tol, sorted_batch):
    from dgll_amd.nn.GlobalPooling import Pooling, maxPooling, meanPooling, sumPooling

    g = torch.Generator().manual_seed(5)
    sizes = [1, 700, 0, 33, 3000, 2, 0]                      # empty graphs, a one-node graph, long segments
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    if not sorted_batch:
        batch = batch[torch.randperm(batch.numel(), generator=g)]
    x = torch.randn(batch.numel(), 50, generator=g)
    xd = x.cuda().to(dtype).requires_grad_(True)
    xr = xd.detach().float().cpu().requires_grad_(True)
    w = torch.randn(len(sizes), 150, generator=g)
    out = Pooling(["sum", "mean", "max"])(xd, batch.cuda(), len(sizes))
    ref = torch.cat([torch_ref.scatter_pool(xr, batch, len(sizes), r) for r in ("sum", "mean", "max")], dim=-1)
    scale = float(ref.detach().abs().max())
    torch.testing.assert_close(out.float().cpu(), ref, rtol=tol, atol=tol * scale)
    (out.float() * w.cuda()).sum().backward()
    (ref * w).sum().backward()
    torch.testing.assert_close(xd.grad.float().cpu(), xr.grad, rtol=tol, atol=tol * float(xr.grad.abs().max()))
    for fn, r in ((sumPooling, "sum"), (meanPooling, "mean"), (maxPooling, "max")):
        torch.testing.assert_close(fn(xd.detach(), batch.cuda()).float().cpu(), torch_ref.scatter_pool(xr.detach(), batch, None, r),
                                   rtol=tol, atol=tol * scale)
    assert sumPooling(xd.detach(), None).shape == (1, 50)


def test_pooling_rejects_cpu_and_bad_ids():
    from dgll_amd.nn.GlobalPooling import segments_of, sumPooling

    with pytest.raises(RuntimeError):
        sumPooling(torch.zeros(4, 8), torch.zeros(4, dtype=torch.long))
    with pytest.raises(ValueError):
        segments_of(torch.tensor([0, 3]).cuda(), size=2)
