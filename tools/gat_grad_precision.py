#!/usr/bin/env python3
"""How far the bf16 GAT passes are from float64 arithmetic ON THE SAME bf16 OPERANDS (h, the incoming gradient), pass by pass:
output, grad_h, grad_s, grad_t of ops.gat_aggregate with and without the ELU epilogue.  The difference left is
what the kernels themselves round (DN rows, stored outputs), nothing upstream."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import ops, synth  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
g_cpu = synth.rmat_graph(12, 12, seed=4, device="cpu", symmetric=True, weighted=False, self_loops=True)
g = g_cpu.to(dev)
n, alpha = g.n_rows, 0.2
row = torch.repeat_interleave(torch.arange(n), g_cpu.rowptr[1:] - g_cpu.rowptr[:-1])
col = g_cpu.col.long()


def reference(h, s, t, gout, heads, elu):
    h, s, t = h.double().requires_grad_(), s.double().requires_grad_(), t.double().requires_grad_()
    fo = h.shape[1] // heads
    outs = []
    for k in range(heads):
        hk = h[:, k * fo:(k + 1) * fo]
        z = torch.nn.functional.leaky_relu(s[:, k][row] + t[:, k][col], alpha)
        e = torch.exp(-z)
        den = torch.zeros(n, dtype=torch.float64).index_add_(0, row, e)
        hp = torch.zeros(n, fo, dtype=torch.float64).index_add_(0, row, e[:, None] * hk[col]) / den[:, None]
        outs.append(torch.nn.functional.elu(hp) if elu else hp)
    out = torch.cat(outs, 1)
    (out * gout.double()).sum().backward()
    return out.detach(), h.grad, s.grad, t.grad


def rel(a, b):
    return float((a.double().cpu() - b).norm() / b.norm())


for heads, fo in ((8, 32), (1, 48)):
    for elu in (False, True):
        h = ops.alloc_features(n, heads * fo, torch.bfloat16, dev, pad_to=64)
        h.copy_(torch.randn(n, heads * fo, device=dev))
        s = torch.randn(n, heads, device=dev)
        t = torch.randn(n, heads, device=dev)
        gout = torch.randn(n, heads * fo, device=dev).to(torch.bfloat16)
        ref = reference(h.float().cpu(), s.cpu(), t.cpu(), gout.float().cpu(), heads, elu)
        hh = h.detach().requires_grad_()
        ss, tt = s.clone().requires_grad_(), t.clone().requires_grad_()
        out = ops.gat_aggregate(g, hh, ss, tt, heads, alpha, apply_elu=elu, mode=0)
        (out.float() * gout.float()).sum().backward()
        print("heads %d x %d elu %d: out %.2e (vs bf16-rounded reference %.2e)  grad_h %.2e  grad_s %.2e  grad_t %.2e" %
              (heads, fo, elu, rel(out, ref[0]), rel(out, ref[0].to(torch.bfloat16).double()), rel(hh.grad, ref[1]),
               rel(ss.grad, ref[2]), rel(tt.grad, ref[3])), flush=True)

print("--- ops.gat_layer (scores formed from h inside the node: grad_h includes the score path)")
for heads, fo in ((8, 32), (1, 48)):
    h = ops.alloc_features(n, heads * fo, torch.bfloat16, dev, pad_to=64)
    h.copy_(torch.randn(n, heads * fo, device=dev))
    A = torch.zeros(heads * fo, 2 * heads, device=dev)
    for k in range(heads):
        A[k * fo:(k + 1) * fo, k] = torch.randn(fo, device=dev) * 0.25
        A[k * fo:(k + 1) * fo, heads + k] = torch.randn(fo, device=dev) * 0.25
    A = A.to(torch.bfloat16)
    gout = torch.randn(n, heads * fo, device=dev).to(torch.bfloat16)
    hd = h.float().cpu().double().requires_grad_()
    Ad = A.float().cpu().double().requires_grad_()
    st = hd @ Ad
    fo_ = fo
    outs = []
    for k in range(heads):
        hk = hd[:, k * fo_:(k + 1) * fo_]
        z = torch.nn.functional.leaky_relu(st[:, k][row] + st[:, heads + k][col], alpha)
        e = torch.exp(-z)
        den = torch.zeros(n, dtype=torch.float64).index_add_(0, row, e)
        outs.append(torch.nn.functional.elu(torch.zeros(n, fo_, dtype=torch.float64).index_add_(0, row, e[:, None] * hk[col]) / den[:, None]))
    (torch.cat(outs, 1) * gout.float().cpu().double()).sum().backward()
    hh = h.detach().requires_grad_()
    AA = A.detach().clone().requires_grad_()
    out = ops.gat_layer(g, hh, AA, heads, alpha, apply_elu=True, pack_scores=True)
    (out.float() * gout.float()).sum().backward()
    mask = (Ad.detach() != 0)
    print("heads %d x %d: grad_h (aggregation + score path) %.2e   grad_A %.2e   share of the score path in grad_h's norm: see grad_s above" %
          (heads, fo, rel(hh.grad, hd.grad), rel(AA.grad.float().cpu().double() * mask, Ad.grad * mask)), flush=True)
