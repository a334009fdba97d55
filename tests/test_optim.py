"""FlatAdam (dgll_amd/optim.py): torch.optim.Adam's arithmetic over one flat buffer, gradient slots that autograd adopts without
a copy, packed-weight bookkeeping.  Host-tensor branch here; the HIP kernel is covered by tests/test_optim_gpu.py."""
import torch

from dgll_amd.optim import FlatAdam, grad_slot_of


class _Lin(torch.autograd.Function):
    """x . W with the weight gradient written into the optimizer's slot -- the pattern of fused_layers / dist."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.w = w
        ctx.save_for_backward(x, w)
        return x @ w

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gw = x.t() @ g
        slot = grad_slot_of(ctx.w)
        if slot is not None:
            slot.copy_(gw)
            gw = slot
        return g @ w.t(), gw


def _models():
    torch.manual_seed(0)
    a = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(5, 7)), torch.nn.Parameter(torch.randn(7, 3)), torch.nn.Parameter(torch.randn(3))])
    b = torch.nn.ParameterList([torch.nn.Parameter(p.detach().clone()) for p in a])
    return a, b


def test_flat_adam_matches_torch_adam_on_host_tensors():
    a, b = _models()
    ref = torch.optim.Adam(a, lr=1e-2, weight_decay=0.01)
    opt = FlatAdam(b, lr=1e-2, weight_decay=0.01, pack_weights=False)
    x = torch.randn(11, 5)
    for _ in range(5):
        for params, o in ((a, ref), (b, opt)):
            o.zero_grad(set_to_none=True)
            y = torch.relu(x @ params[0]) @ params[1] + params[2]
            y.square().mean().backward()
            o.step()
    for p, q in zip(a, b):
        torch.testing.assert_close(q, p, rtol=1e-6, atol=1e-7)
    # parameters are views of ONE buffer and so are their gradients
    assert all(q.data_ptr() == opt.flat.data_ptr() + 4 * o for q, o in zip(b, opt.offsets))
    assert all(q.grad.data_ptr() == opt.grad.data_ptr() + 4 * o for q, o in zip(b, opt.offsets))


def test_gradient_slot_is_adopted_without_a_copy_and_claimed_once():
    _, b = _models()
    opt = FlatAdam(b, lr=1e-2, pack_weights=False)
    x = torch.randn(4, 5)
    opt.zero_grad()
    _Lin.apply(x, b[0]).sum().backward()
    assert b[0].grad.data_ptr() == opt.grad_slot(b[0]).data_ptr()          # autograd took the slot view itself
    want = x.t() @ torch.ones(4, 7)
    torch.testing.assert_close(b[0].grad, want)
    # a second backward without zero_grad accumulates (the slot is not handed out twice)
    _Lin.apply(x, b[0]).sum().backward()
    torch.testing.assert_close(b[0].grad, 2 * want)
    # the same parameter used by two nodes of one graph: one claims the slot, the other's gradient is added to it
    opt.zero_grad()
    (_Lin.apply(x, b[0]).sum() + 3.0 * _Lin.apply(x, b[0]).sum()).backward()
    torch.testing.assert_close(b[0].grad, 4 * want)
    opt.step()                                                              # unused parameters get a zero gradient
    assert float(opt.grad_slot(b[1]).abs().sum()) == 0.0


def test_state_dict_round_trip():
    _, b = _models()
    opt = FlatAdam(b, pack_weights=False)
    opt.zero_grad()
    (b[0].sum() + b[2].sum()).backward()
    opt.step()
    sd = opt.state_dict()
    _, c = _models()
    opt2 = FlatAdam(c, pack_weights=False)
    opt2.load_state_dict(sd)
    assert opt2.steps == 1 and torch.equal(opt2.exp_avg, opt.exp_avg)
