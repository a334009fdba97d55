"""hipGraph capture of a full-batch training step (dgll_amd/graphs.py): replays must train exactly like the eager loop."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_graphed_step_equals_eager(cuda_device):
    from dgll_amd.evaluation.ppi import GCN
    from dgll_amd.graphs import GraphedTrainStep

    torch.manual_seed(3)
    n = 600
    e = torch.randint(0, n, (2, 9000), device=cuda_device)
    e = e[:, e[0] != e[1]]
    x = torch.randn(n, 50, device=cuda_device)
    y = (torch.rand(n, 12, device=cuda_device) < 0.3).float()
    base = GCN(50, 64, 12, 2).to(cuda_device)
    with torch.no_grad():
        for layer in base.layers:
            layer.weight.mul_(0.1)
    crit = torch.nn.CrossEntropyLoss()

    def train(model, graphed, steps):
        opt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=True)
        losses = []
        if graphed:
            step = GraphedTrainStep(lambda: crit(model(e, x), y), opt, warmup=2)     # 2 warm-up steps are real steps
            for _ in range(steps - 2):
                losses.append(float(step()))
        else:
            for i in range(steps):
                opt.zero_grad(set_to_none=True)
                loss = crit(model(e, x), y)
                loss.backward()
                opt.step()
                if i >= 2:
                    losses.append(float(loss.detach()))
        return losses, [p.detach().clone() for p in model.parameters()]

    la, pa = train(copy.deepcopy(base), False, 8)
    lb, pb = train(copy.deepcopy(base), True, 8)
    assert la[-1] < la[0]                                     # it trains
    torch.testing.assert_close(torch.tensor(lb), torch.tensor(la), rtol=1e-5, atol=1e-6)
    for a, b in zip(pa, pb):
        torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-6)


def test_whole_epoch_in_one_graph(cuda_device):
    """The PPI loop: the steps over all training graphs, updating the same parameters and Adam state, in ONE HIP graph."""
    from dgll_amd.evaluation.ppi import GCN
    from dgll_amd.graphs import GraphedTrainStep

    torch.manual_seed(5)
    data = []
    for n in (300, 2450, 3380):                                    # PPI-sized: the split-K weight-gradient path too
        e = torch.randint(0, n, (2, 14 * n), device=cuda_device)
        e = torch.cat([e, e.flip(0)], dim=1)
        data.append((e[:, e[0] != e[1]], torch.randn(n, 50, device=cuda_device), (torch.rand(n, 121, device=cuda_device) < 0.3).float()))
    base = GCN(50, 64, 121, 2).to(cuda_device)
    with torch.no_grad():
        for layer in base.layers:
            layer.weight.mul_(0.1)
    crit = torch.nn.CrossEntropyLoss()
    epochs = 6

    eager = copy.deepcopy(base)
    opt = torch.optim.Adam(eager.parameters(), lr=0.01, capturable=True)
    ref = []
    for _ in range(epochs):
        for e, x, y in data:
            opt.zero_grad(set_to_none=True)
            loss = crit(eager(e, x), y)
            loss.backward()
            opt.step()
            ref.append(float(loss.detach()))

    model = copy.deepcopy(base)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=True)
    # the whole epoch (one step per training graph) as ONE HIP graph; epoch 0 = its warm-up pass
    epoch = GraphedTrainStep([lambda e=e, x=x, y=y: crit(model(e, x), y) for e, x, y in data], opt, warmup=1)
    got = []
    for _ in range(epochs - 1):
        epoch()
        vals = [float(v) for v in epoch.losses]                     # .item() reads only (see GraphedTrainStep.__call__)
        assert abs(float(epoch.total) - sum(vals)) <= 1e-4 * abs(sum(vals))
        got += vals
    torch.testing.assert_close(torch.tensor(got), torch.tensor(ref[len(data):]), rtol=1e-4, atol=1e-5)
    for a, b in zip(eager.parameters(), model.parameters()):
        torch.testing.assert_close(b, a, rtol=1e-4, atol=1e-5)


def test_graphed_step_requires_capturable_optimizer(cuda_device):
    from dgll_amd.graphs import GraphedTrainStep

    w = torch.nn.Parameter(torch.ones(4, device=cuda_device))
    with pytest.raises(ValueError):
        GraphedTrainStep(lambda: (w * w).sum(), torch.optim.Adam([w], capturable=False))


def test_graphed_bf16_sage_step_uses_the_mfma_and_gate_kernels(cuda_device):
    """Full-graph GraphSAGE (bf16: MFMA transforms, gated SpMM epilogues, the loss kernel) captured in a HIP graph."""
    from dgll_amd import nn as dnn
    from dgll_amd import ops, synth
    from dgll_amd.graphs import GraphedTrainStep

    torch.manual_seed(0)
    g = synth.products_like_graph(cuda_device, seed=1, n=20000, n_undirected=300000, locality=0.9)
    n = g.n_rows
    x = ops.alloc_features(n, 100, torch.bfloat16, cuda_device, pad_to=64)
    x.copy_(torch.randn(n, 100, device=cuda_device))
    labels = torch.randint(0, 47, (n,), device=cuda_device)
    base = dnn.GraphSage(100, [256, 256, 47], None).to(cuda_device)

    def run(graphed):
        model = copy.deepcopy(base)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)
        fn = lambda: ops.cross_entropy(model.forward_graph(g, x), labels)      # noqa: E731
        losses = []
        if graphed:
            step = GraphedTrainStep(fn, opt, warmup=2)
            for _ in range(4):
                losses.append(float(step()))
        else:
            for i in range(6):
                opt.zero_grad(set_to_none=True)
                loss = fn()
                loss.backward()
                opt.step()
                if i >= 2:
                    losses.append(float(loss.detach()))
        return losses, [p.detach().clone() for p in model.parameters()]

    la, pa = run(False)
    lb, pb = run(True)
    torch.testing.assert_close(torch.tensor(lb), torch.tensor(la), rtol=1e-4, atol=1e-4)
    for a, b in zip(pa, pb):
        torch.testing.assert_close(b, a, rtol=1e-3, atol=1e-5)
