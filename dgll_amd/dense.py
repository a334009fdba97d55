"""Dense transforms that surround the aggregation (x.W of gcnconv.py:30, sageconv.py:41,72, gatconv.py:31,117).

Forward and input-gradient products are plain library GEMMs (hipBLASLt through torch.mm: tall-skinny
[N, F].[F, H], already within ~25 % of the HBM bound).  The WEIGHT gradient x^T.g reduces over millions of rows into a
tiny [F, H] output; the library runs that as a handful of workgroups (3.6 ms at N = 2.4 M, F = H = 256 on MI355X), so
it is computed split-K: the row dimension is cut into slabs that are multiplied as one batched GEMM and summed
(0.6 ms).
"""
import torch

_SLABS = 256


def grad_weight(x, g):
    """x^T . g  for x [M, K], g [M, N] -> fp32 [K, N], split over M."""
    m = x.shape[0]
    if m < 64 * _SLABS:
        return torch.mm(x.t(), g).float()
    rows = m // _SLABS
    main = rows * _SLABS
    xs = x[:main].view(_SLABS, rows, x.shape[1])
    gs = g[:main].view(_SLABS, rows, g.shape[1])
    out = torch.bmm(xs.transpose(1, 2), gs).float().sum(0)
    if main < m:
        out += torch.mm(x[main:].t(), g[main:]).float()
    return out


class _SageTransform(torch.autograd.Function):
    """relu?( h.Ws + agg.Wn ) with one fused add (addmm) and in-place activation; backward shares the masked gradient
    between the four products (sageconv.py:71-82 computes the same terms as separate autograd nodes)."""

    @staticmethod
    def forward(ctx, h, agg, ws, wn, relu):
        wsd, wnd = ws.to(h.dtype), wn.to(h.dtype)
        out = torch.addmm(torch.mm(h, wsd), agg, wnd)
        if relu:
            out.relu_()
        ctx.relu = relu
        ctx.save_for_backward(h, agg, wsd, wnd, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, g):
        h, agg, wsd, wnd, out = ctx.saved_tensors
        if ctx.relu:
            g = torch.ops.aten.threshold_backward(g.contiguous(), out, 0)   # one vectorised pass: g where out > 0
        g = g.contiguous()
        gh = torch.mm(g, wsd.t()) if ctx.needs_input_grad[0] else None
        gagg = torch.mm(g, wnd.t()) if ctx.needs_input_grad[1] else None
        gws = grad_weight(h, g) if ctx.needs_input_grad[2] else None
        gwn = grad_weight(agg, g) if ctx.needs_input_grad[3] else None
        return gh, gagg, gws, gwn, None


def sage_transform(h, agg, ws, wn, relu):
    return _SageTransform.apply(h, agg, ws, wn, relu)
