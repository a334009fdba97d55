"""ops.row_slices: overlapping row ranges of one matrix through ONE autograd node (the stacked hops of a sampled GraphSAGE layer's
input, sageconv.py:103-114's pyramid) -- same values and the same gradient as plain slicing."""
import pytest
import torch


@pytest.mark.parametrize("bounds", [[(0, 8), (5, 8), (8, 20)], [(0, 5), (5, 12)], [(3, 9), (0, 20), (7, 15)], [(2, 4)], [(0, 20), (0, 20)],
                                    [(4, 4), (10, 20)]])
def test_row_slices_match_plain_slicing(bounds):
    from dgll_amd import ops

    torch.manual_seed(len(bounds))
    x = torch.randn(20, 3, requires_grad=True)
    views = ops.row_slices(x, bounds)
    assert [tuple(v.shape) for v in views] == [(b - a, 3) for a, b in bounds]
    for v, (a, b) in zip(views, bounds):
        assert torch.equal(v, x[a:b])
    w = [torch.randn_like(v) for v in views]
    g, = torch.autograd.grad(sum((v * wi).sum() for v, wi in zip(views, w)), x)
    xr = x.detach().clone().requires_grad_()
    gr, = torch.autograd.grad(sum((xr[a:b] * wi).sum() for (a, b), wi in zip(bounds, w)), xr)
    torch.testing.assert_close(g, gr, rtol=1e-6, atol=1e-6)
    # a range that receives no gradient, and an input that needs none
    views = ops.row_slices(x, bounds)
    g2, = torch.autograd.grad((views[-1] * w[-1]).sum(), x)
    a, b = bounds[-1]
    ref = torch.zeros_like(x)
    ref[a:b] = w[-1]
    torch.testing.assert_close(g2, ref)
    assert all(not v.requires_grad for v in ops.row_slices(x.detach(), bounds))
