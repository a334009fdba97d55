"""Whole-layer autograd nodes for full-graph GraphSAGE (forward_graph): the layer's input feeds both the self term
and the neighbour aggregation, so its gradient has two contributions.  Written as separate autograd nodes (as the
reference's op-by-op layers are, sageconv.py:70-83) they are summed by an extra elementwise pass over [N, F]; here the
second contribution is accumulated by the epilogue of the kernel that produces it (SpMM `accumulate`, or addmm)."""
import torch

from . import dense, ops


def _aligned(t):
    return (t.stride(0) * t.element_size()) % 16 == 0 and t.data_ptr() % 16 == 0 and t.stride(1) == 1


class _SageGraphLayer(torch.autograd.Function):
    """out = act(h.Ws + reduce_A(h).Wn)  -- aggregate-then-transform (sageconv.py:33-41,72-75)."""

    @staticmethod
    def forward(ctx, h, ws, wn, graph, reduce, relu):
        agg = ops.spmm_raw(graph, h, reduce=reduce)
        wsd, wnd = ws.to(h.dtype), wn.to(h.dtype)
        if dense._mfma_ok(h, agg) and ws.shape[1] <= 256:
            out = dense.transform_bf16(h, wsd.t(), agg, wnd.t(), relu=relu)
        else:
            out = torch.addmm(torch.mm(h, wsd), agg, wnd)
            if relu:
                out.relu_()
        ctx.graph, ctx.reduce, ctx.relu = graph, reduce, relu
        ctx.save_for_backward(h, agg, wsd, wnd, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, g):
        h, agg, wsd, wnd, out = ctx.saved_tensors
        graph = ctx.graph
        g = g.contiguous()
        if ctx.relu:
            g = torch.ops.aten.threshold_backward(g, out, 0)
        gws = dense.grad_weight(h, g) if ctx.needs_input_grad[1] else None
        gwn = dense.grad_weight(agg, g) if ctx.needs_input_grad[2] else None
        gh = None
        if ctx.needs_input_grad[0]:
            gh = torch.mm(g, wsd.t())                      # self path
            gagg = torch.mm(g, wnd.t())
            gt, _ = graph.transpose()
            tval = gt.val
            if ctx.reduce == "mean":
                scale = graph.mean_scale_transposed()
                tval = scale if tval is None else tval * scale
            if _aligned(gh) and _aligned(gagg):            # neighbour path lands on top of the self path in the epilogue
                ops.spmm_raw(gt, gagg, val=tval, reduce="sum", out=gh, accumulate=True)
            else:
                gh = gh + ops.spmm_raw(gt, gagg, val=tval, reduce="sum")
        return gh, gws, gwn, None, None, None


class _SageGraphLayerTransformFirst(torch.autograd.Function):
    """out = act(h.Ws + reduce_A(h.Wn))  -- the narrowing layer: the NARROW product is aggregated (mean/sum are linear)."""

    @staticmethod
    def forward(ctx, h, ws, wn, graph, reduce, relu):
        wsd, wnd = ws.to(h.dtype), wn.to(h.dtype)
        # the narrow product is gathered next: one 128-byte line per row (ld_align) instead of rows straddling two lines
        z = (dense.transform_bf16(h, wnd.t(), ld_align=64 if wn.shape[1] < 64 else None)
             if (dense._mfma_ok(h) and wn.shape[1] <= 256) else torch.mm(h, wnd))
        out = torch.addmm(ops.spmm_raw(graph, z, reduce=reduce), h, wsd)
        if relu:
            out.relu_()
        ctx.graph, ctx.reduce, ctx.relu = graph, reduce, relu
        ctx.save_for_backward(h, wsd, wnd, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, g):
        h, wsd, wnd, out = ctx.saved_tensors
        graph = ctx.graph
        g = g.contiguous()
        if ctx.relu:
            g = torch.ops.aten.threshold_backward(g, out, 0)
        # d/dz of reduce_A(z): A^T . g (1/deg folded into the padded copy of the narrow gradient)
        line = 128 // g.element_size()
        gp = ops.alloc_features(g.shape[0], g.shape[1], g.dtype, g.device, pad_to=line if g.shape[1] < line else (16 // g.element_size()))
        if ctx.reduce == "mean":
            torch.mul(g, (1.0 / graph.degrees().clamp(min=1).to(torch.float32)).unsqueeze(1).to(g.dtype), out=gp)
        else:
            gp.copy_(g)
        gt, _ = graph.transpose()
        gz = ops.spmm_raw(gt, gp, val=gt.val, reduce="sum")
        gws = dense.grad_weight(h, g) if ctx.needs_input_grad[1] else None
        gwn = dense.grad_weight(h, gz) if ctx.needs_input_grad[2] else None
        gh = None
        if ctx.needs_input_grad[0]:
            gh = torch.addmm(torch.mm(g, wsd.t()), gz, wnd.t())     # both paths in one GEMM epilogue
        return gh, gws, gwn, None, None, None


def sage_graph_layer(layer, graph, h):
    """Full-graph sageConv (x_dst is x_src) through the fused nodes when the layer has the standard configuration;
    None otherwise (the caller falls back to forward_block)."""
    from . import backend as F

    if not (h.is_cuda and layer.aggr_hid_method == "sum" and not layer.neighborAgg.use_bias
            and layer.aggr_neighbor_method in ("mean", "sum") and layer.activation in (None, F.relu)):
        return None
    relu = layer.activation is not None
    if layer.transform_first(h):
        return _SageGraphLayerTransformFirst.apply(h, layer.weight, layer.neighborAgg.weight, graph,
                                                   layer.aggr_neighbor_method, relu)
    return _SageGraphLayer.apply(h, layer.weight, layer.neighborAgg.weight, graph, layer.aggr_neighbor_method, relu)
