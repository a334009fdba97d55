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
    before = b[1].detach().clone()
    opt.step()                                                              # a parameter without gradient is skipped, as torch.optim.Adam does
    assert b[1].grad is None and torch.equal(b[1], before) and opt.param_steps == [1, 0, 0]


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


def test_parameters_without_a_gradient_are_skipped_exactly_like_torch_adam():
    """FlatAdam zero-filled the slot of a parameter without gradient and updated the whole buffer -- stale momentum
    kept moving it, both moments decayed, weight decay was applied.  torch.optim.Adam skips such a parameter entirely (its own step
    count too).  Here: the middle parameter is unused in steps 2 and 3 of 6, weight_decay > 0."""
    a, b = _models()
    ref = torch.optim.Adam(a, lr=1e-2, weight_decay=0.05)
    opt = FlatAdam(b, lr=1e-2, weight_decay=0.05, pack_weights=False)
    x = torch.randn(11, 5, generator=torch.Generator().manual_seed(3))
    for it in range(6):
        for params, o in ((a, ref), (b, opt)):
            o.zero_grad(set_to_none=True)
            h = torch.relu(x @ params[0])
            y = (h @ params[1] + params[2]) if it not in (2, 3) else h.sum(1, keepdim=True) + params[2]
            y.square().mean().backward()
            o.step()
        assert (b[1].grad is None) == (it in (2, 3))
    for p, q in zip(a, b):
        torch.testing.assert_close(q, p, rtol=1e-6, atol=1e-7)
    assert opt.param_steps == [6, 4, 6] and opt.steps == 6
    assert [int(ref.state[p]["step"]) for p in a] == opt.param_steps


def test_param_groups_lr_schedulers_one_shot_grad_scale_and_clipping():
    a, b = _models()
    ref = torch.optim.Adam([{"params": [a[0]], "lr": 3e-2}, {"params": [a[1], a[2]]}], lr=1e-2)
    opt = FlatAdam([{"params": [b[0]], "lr": 3e-2}, {"params": [b[1], b[2]]}], lr=1e-2, pack_weights=False)
    assert isinstance(opt, torch.optim.Optimizer) and len(opt.param_groups) == 2 and opt.param_groups[0]["lr"] == 3e-2
    sched_a = torch.optim.lr_scheduler.StepLR(ref, step_size=2, gamma=0.5)
    sched_b = torch.optim.lr_scheduler.StepLR(opt, step_size=2, gamma=0.5)          # reads and writes param_groups
    x = torch.randn(9, 5, generator=torch.Generator().manual_seed(4))
    for _ in range(5):
        for params, o, sch in ((a, ref, sched_a), (b, opt, sched_b)):
            o.zero_grad(set_to_none=True)
            (torch.relu(x @ params[0]) @ params[1] + params[2]).square().mean().backward()
            o.step()
            sch.step()
    for p, q in zip(a, b):
        torch.testing.assert_close(q, p, rtol=1e-6, atol=1e-7)
    assert opt.param_groups[0]["lr"] == ref.param_groups[0]["lr"] == 3e-2 * 0.25
    # grad_scale is consumed by the step that uses it (a later local step is not silently averaged)
    opt.zero_grad()
    (b[0].sum() * 2.0).backward()
    opt.grad_scale = 0.5
    assert abs(float(opt.grad_norm()) - 0.5 * float(b[0].grad.norm())) < 1e-5
    opt.step()
    assert opt.grad_scale == 1.0
    # clipping: the coefficient rides the one-shot scale; equals torch's clip + step
    a2, b2 = _models()
    r2, o2 = torch.optim.Adam(a2, lr=1e-2), FlatAdam(b2, lr=1e-2, pack_weights=False)
    for params, o in ((a2, r2), (b2, o2)):
        o.zero_grad(set_to_none=True)
        ((torch.relu(x @ params[0]) @ params[1] + params[2]) * 50.0).square().mean().backward()
    n_ref = torch.nn.utils.clip_grad_norm_(list(a2), 1.0)
    n_flat = o2.clip_grad_norm_(1.0)
    assert abs(float(n_ref) - float(n_flat)) < 1e-3 * float(n_ref)
    r2.step()
    o2.step()
    for p, q in zip(a2, b2):
        torch.testing.assert_close(q, p, rtol=1e-5, atol=1e-6)
    # hyper-parameters travel with the state dict
    sd = opt.state_dict()
    _, c = _models()
    fresh = FlatAdam([{"params": [c[0]]}, {"params": [c[1], c[2]]}], lr=1.0, pack_weights=False)
    fresh.load_state_dict(sd)
    assert fresh.param_groups[0]["lr"] == opt.param_groups[0]["lr"] and fresh.param_steps == opt.param_steps
