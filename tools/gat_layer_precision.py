#!/usr/bin/env python3
"""One SpGAT layer in bf16 (transform -> scores -> fused aggregation -> ELU) against float64 with the bf16 storage points emulated:
where do the parameter gradients' ~1e-2 come from?  Prints the error of every intermediate gradient."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import dense, ops, synth  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
g_cpu = synth.rmat_graph(12, 12, seed=4, device="cpu", symmetric=True, weighted=False, self_loops=True)
g = g_cpu.to(dev)
n, alpha = g.n_rows, 0.2
row = torch.repeat_interleave(torch.arange(n), g_cpu.rowptr[1:] - g_cpu.rowptr[:-1])
col = g_cpu.col.long()
rnd = lambda t: t.to(torch.bfloat16).to(t.dtype)       # noqa: E731


class Store(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return rnd(t)

    @staticmethod
    def backward(ctx, gr):
        return rnd(gr)


def rel(a, b):
    return float((a.double().cpu() - b).norm() / b.norm())


for fin, heads, fo, fo_pad in ((100, 8, 32, 32), (256, 1, 47, 48)):
    x = ops.alloc_features(n, fin, torch.bfloat16, dev)
    x.copy_(torch.randn(n, fin, device=dev))
    W = (torch.randn(fin, heads * fo, device=dev) * (1.414 * (2.0 / (fin + fo)) ** 0.5)).requires_grad_()
    a = (torch.randn(heads, 2 * fo, device=dev) * (1.414 * (2.0 / (1 + 2 * fo)) ** 0.5)).requires_grad_()
    gout = torch.randn(n, heads * fo, device=dev).to(torch.bfloat16)

    # ---- float64 with storage emulation
    xd = x.float().cpu().double().requires_grad_()
    Wd = rnd(W.detach().cpu()).double().requires_grad_()
    ad = rnd(a.detach().cpu()).double().requires_grad_()
    hd = Store.apply(xd @ Wd)
    hd.retain_grad()
    outs = []
    for k in range(heads):
        hk = hd[:, k * fo:(k + 1) * fo]
        z = torch.nn.functional.leaky_relu((hk @ ad[k, :fo])[row] + (hk @ ad[k, fo:])[col], alpha)
        e = torch.exp(-z)
        den = torch.zeros(n, dtype=torch.float64).index_add_(0, row, e)
        outs.append(torch.nn.functional.elu(torch.zeros(n, fo, dtype=torch.float64).index_add_(0, row, e[:, None] * hk[col]) / den[:, None]))
    od = torch.cat(outs, 1)
    (od * gout.float().cpu().double()).sum().backward()

    # ---- the GPU path, as gatconv._fused_heads builds it
    Wp = W if fo_pad == fo else torch.cat([torch.nn.functional.pad(W[:, k * fo:(k + 1) * fo], (0, fo_pad - fo)) for k in range(heads)], 1)
    xg = x.detach().requires_grad_()
    h = dense.linear(xg, Wp)
    h.retain_grad()
    A = h.new_zeros(heads * fo_pad, 2 * heads)
    for k in range(heads):
        A[k * fo_pad:k * fo_pad + fo, k] = a[k, :fo].to(h.dtype)
        A[k * fo_pad:k * fo_pad + fo, heads + k] = a[k, fo:].to(h.dtype)
    out = ops.gat_layer(g, h, A, heads, alpha, apply_elu=True, pack_scores=True)
    outc = torch.cat([out[:, k * fo_pad:k * fo_pad + fo] for k in range(heads)], 1)
    (outc.float() * gout.float()).sum().backward()
    hg = torch.cat([h.grad[:, k * fo_pad:k * fo_pad + fo] for k in range(heads)], 1)
    print("Fin %d, %d heads x %d: out %.2e  grad_h %.2e  grad_W %.2e  grad_a %.2e  grad_x %.2e   (|grad_h| rms %.3e)" %
          (fin, heads, fo, rel(outc, od.detach()), rel(hg, hd.grad), rel(W.grad, Wd.grad), rel(a.grad, ad.grad), rel(xg.grad, xd.grad),
           float(hd.grad.pow(2).mean().sqrt())), flush=True)
    # the same weight gradient from the REFERENCE's grad_h rounded to bf16: what storing grad_h in bf16 costs by itself
    gw_from_rounded = x.float().cpu().double().t() @ rnd(hd.grad.float()).double()
    print("    grad_W formed from the float64 grad_h rounded to bf16: %.2e" % rel(gw_from_rounded, Wd.grad), flush=True)
