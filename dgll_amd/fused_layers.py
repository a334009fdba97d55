"""Whole-layer autograd nodes for full-graph GraphSAGE (forward_graph): the layer's input feeds both the self term
and the neighbour aggregation, so its gradient has two contributions.  Written as separate autograd nodes (as the
reference's op-by-op layers are, sageconv.py:70-83) they are summed by an extra elementwise pass over [N, F]; here the
second contribution is accumulated by the epilogue of the kernel that produces it (SpMM `accumulate`, or addmm).

ReLU backward between two such nodes is fused the same way.  Inside GraphSage.forward_graph the activations between
layers are seen by nobody else, so the pair (producer, consumer) may agree that the consumer hands back the gradient
ALREADY masked by (h > 0) -- written by the epilogue of the kernel that produces it (`gate_input`: SpMM gate /
MFMA output gate) -- and the producer skips its own masking pass (`grad_is_gated`).  Masking is linear and idempotent, so
the parameter gradients and the model-input gradient are exactly those of the unfused graph."""
import os

import torch

from . import dense, ops
from .optim import grad_slot_of


# The fused aggregate -> transform kernel (csrc/fused_sage.hip: gather a 64-row tile into LDS, MFMA it against weights streamed
# from L2) is correct and tested, but MEASURED SLOWER than the two launches it replaces on MI355X (products-sized graph, F = 256,
# tools/fused_probe.py): 6.70 ms against SpMM 4.35 + transform 0.88 ms; without the self operand (the reference kernel's
# relu(A.X.W) shape) 5.70 ms.  The gather needs every wavefront slot and register the CU has to keep enough loads in flight,
# and whatever the tile does besides gathering (sixteen dependent L2 round trips for the weight fragments, under a memory
# system the gathers keep saturated) comes straight out of that.  So the layers run SpMM + MFMA transform; set True to route
# them through the fused launch.
FUSE_AGGREGATE_TRANSFORM = False


# Backward of the aggregate-first layer, input gradient  g.Ws^T + A^T(scale . (g.Wn^T)).  Two orders (the mean is linear):
#   "transform-first"  [gh | gagg] = g.[Ws^T | Wn^T] (one dual MFMA launch), then the transposed SpMM of gagg ACCUMULATES onto gh and
#                      applies the ReLU mask of the layer below in its epilogue (rounds 2-4);
#   "aggregate-first"  gt = A^T(scale . g) (plain weighted SpMM: no read-modify-write of the output, no gate operand, the 8-wavefront
#                      instantiation), then ONE two-operand MFMA launch  gate(g.Ws^T + gt.Wn^T)  -- when g is no wider than the layer's input.
# dgll_amd.fused_layers.BACKWARD_ORDER = "auto" picks aggregate-first whenever it applies; measured in bench.py's step (CHANGELOG).
BACKWARD_ORDER = __import__("os").environ.get("DGLL_BACKWARD_ORDER", "auto")      # auto | transform-first


GATE_BITS = os.environ.get("DGLL_GATE_BITS", "1") != "0"      # 0: the ReLU gates are read back as bf16 activations (A/B, tests)


def _tag_bits(out, bits):
    """The sign bits of an activation travel with the tensor that holds it: (bits, data_ptr, version) -- a consumer uses them only
    for the very storage and version they were written for (an in-place edit in between bumps the version)."""
    out._dgll_gate_bits = (bits, out.data_ptr(), out._version)


def _bits_of(h):
    tag = getattr(h, "_dgll_gate_bits", None)
    if tag is None or not GATE_BITS:
        return None
    bits, ptr, version = tag
    if ptr != h.data_ptr() or version != h._version or bits.shape != (h.shape[0], dense.bit_words(h.shape[1])):
        return None
    return bits


def _aligned(t):
    return (t.stride(0) * t.element_size()) % 16 == 0 and t.data_ptr() % 16 == 0 and t.stride(1) == 1


class _SageGraphLayer(torch.autograd.Function):
    """out = act(h.Ws + reduce_A(h).Wn)  -- aggregate-then-transform (sageconv.py:33-41,72-75)."""

    @staticmethod
    def forward(ctx, h, ws, wn, graph, reduce, relu, grad_is_gated=False, gate_input=False, token=None, bits_box=None):
        wsd, wnd = dense.wcast(ws, h), dense.wcast(wn, h)
        ctx.grad_is_gated, ctx.gate_input, ctx.token = grad_is_gated, gate_input, token
        ctx.h_bits = _bits_of(h) if gate_input else None
        if FUSE_AGGREGATE_TRANSFORM and dense.fused_ok(graph, h, ws.shape[1], h):
            # ONE launch: a workgroup aggregates a 32-row tile into LDS and feeds it to the MFMAs; the aggregated rows are
            # written (the weight gradient needs them) but never read back (csrc/fused_sage.hip)
            out, agg = dense.sage_fused_forward(graph, h, reduce, h, wsd.t(), wnd.t(), relu, keep_agg=True)
        else:
            agg = ops.spmm_raw(graph, h, reduce=reduce)
            if dense._mfma_ok(h, agg) and ws.shape[1] <= 256:
                if relu and bits_box is not None and GATE_BITS:     # the sign bits ride out with the activation (stores only)
                    out, bits = dense.transform_bf16(h, wsd.t(), agg, wnd.t(), relu=relu, bits_out=True)
                    bits_box.append(bits)
                else:
                    out = dense.transform_bf16(h, wsd.t(), agg, wnd.t(), relu=relu)
            else:
                out = dense.mm2_nt(h, wsd.t(), agg, wnd.t(), relu=relu)
        ctx.graph, ctx.reduce, ctx.relu = graph, reduce, relu
        ctx.wparams = (ws, wn)
        ctx.save_for_backward(h, agg, wsd, wnd, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, g):
        h, agg, wsd, wnd, out = ctx.saved_tensors
        graph = ctx.graph
        g = g.contiguous()
        if ctx.relu and not ctx.grad_is_gated and not (ctx.token is not None and ctx.token.masked):
            g = torch.ops.aten.threshold_backward(g, out, 0)
        if ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
            # one launch: g read once for both products; written into the optimizer's gradient slots when it owns the parameters
            gws, gwn = dense.grad_weight_pair(h, agg, g, out1=grad_slot_of(ctx.wparams[0]), out2=grad_slot_of(ctx.wparams[1]))
        else:
            gws = dense.grad_weight(h, g, out=grad_slot_of(ctx.wparams[0])) if ctx.needs_input_grad[1] else None
            gwn = dense.grad_weight(agg, g, out=grad_slot_of(ctx.wparams[1])) if ctx.needs_input_grad[2] else None
        gh = None
        f32 = g.is_cuda and g.dtype == torch.float32 and h.dtype == torch.float32      # the reference's own arithmetic (dgll/__init__.py:1)
        agg_first = (BACKWARD_ORDER != "transform-first" and ctx.needs_input_grad[0] and wsd.shape[1] <= wsd.shape[0]
                     and (f32 or (dense._mfma_ok(g) and wsd.shape[0] <= 256 and _aligned(g)))
                     and (not ctx.gate_input or h.stride(1) == 1))
        if agg_first:
            gt, _ = graph.transpose()
            tval = gt.val
            if ctx.reduce == "mean":
                scale = graph.mean_scale_transposed()
                tval = scale if tval is None else tval * scale
            gtg = ops.spmm_raw(gt, g, val=tval, reduce="sum")                       # A^T (scale . g): plain weighted gather
            if f32:       # both products in one fp32 accumulation, the mask applied by the epilogue (dgll_hip_mm2_f32)
                gh = dense.mm2_nt(g, wsd, gtg, wnd, gate=h if ctx.gate_input else None)
            else:
                gh = dense.transform_bf16(g, wsd, gtg, wnd, out_gate=h if ctx.gate_input else None, gate_bits=ctx.h_bits)
        elif ctx.needs_input_grad[0]:
            gh, gagg = dense.input_grads(g, wsd, wnd)      # self path, neighbour path: one MFMA launch, g read once
            gt, _ = graph.transpose()
            tval = gt.val
            # (folding 1/deg into this product's rows -- transform_bf16(row_scale=) -- so that the SpMM runs unweighted was
            # measured: SpMM 5.21 -> 5.12 ms, but the transform got slower by more: no net gain)
            if ctx.reduce == "mean":
                scale = graph.mean_scale_transposed()
                tval = scale if tval is None else tval * scale
            if _aligned(gh) and _aligned(gagg):            # neighbour path lands on top of the self path in the epilogue
                gate = h if (ctx.gate_input and h.stride(1) == 1) else None
                ops.spmm_raw(gt, gagg, val=tval, reduce="sum", out=gh, accumulate=True, gate=gate)
                if ctx.gate_input and gate is None:
                    gh = torch.ops.aten.threshold_backward(gh, h, 0)
            else:
                gh = gh + ops.spmm_raw(gt, gagg, val=tval, reduce="sum")
                if ctx.gate_input:
                    gh = torch.ops.aten.threshold_backward(gh, h, 0)
        return gh, gws, gwn, None, None, None, None, None, None, None


class _SageGraphLayerTransformFirst(torch.autograd.Function):
    """out = act(h.Ws + reduce_A(h.Wn))  -- the narrowing layer: the NARROW product is aggregated (mean/sum are linear)."""

    @staticmethod
    def forward(ctx, h, ws, wn, graph, reduce, relu, grad_is_gated=False, gate_input=False, token=None, bits_box=None):
        wsd, wnd = dense.wcast(ws, h), dense.wcast(wn, h)
        ctx.grad_is_gated, ctx.gate_input, ctx.token = grad_is_gated, gate_input, token
        ctx.h_bits = _bits_of(h) if gate_input else None
        # the narrow product is gathered next: one 128-byte line per row (ld_align) instead of rows straddling two lines
        z = (dense.transform_bf16(h, wnd.t(), ld_align=64 if wn.shape[1] < 64 else None)
             if (dense._mfma_ok(h) and wn.shape[1] <= 256) else dense.mm_nt(h, wnd.t()))
        if FUSE_AGGREGATE_TRANSFORM and wn.shape[1] == ws.shape[1] and dense.fused_ok(graph, z, ws.shape[1], h):
            # act(reduce_A(z) + h.Ws) in one launch: the aggregate of the narrow product never leaves the workgroup
            out, _ = dense.sage_fused_forward(graph, z, reduce, h, wsd.t(), None, relu, ld_align=64 if ws.shape[1] < 64 else None)
            ctx.graph, ctx.reduce, ctx.relu = graph, reduce, relu
            ctx.wparams = (ws, wn)
            ctx.save_for_backward(h, wsd, wnd, out if relu else None)
            return out
        aggz = ops.spmm_raw(graph, z, reduce=reduce)
        if dense._mfma_ok(h) and aggz.dtype == torch.bfloat16 and aggz.stride(1) == 1 and ws.shape[1] <= 256:
            # act(agg + h.Ws) in ONE MFMA launch: the aggregated term rides in the epilogue (library: addmm + ReLU pass)
            out = dense.transform_bf16(h, wsd.t(), relu=relu, addend=aggz, ld_align=64 if ws.shape[1] < 64 else None)
        else:
            out = dense.mm_nt(h, wsd.t(), relu=relu, addend=aggz)
        ctx.graph, ctx.reduce, ctx.relu = graph, reduce, relu
        ctx.wparams = (ws, wn)
        ctx.save_for_backward(h, wsd, wnd, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, g):
        h, wsd, wnd, out = ctx.saved_tensors
        graph = ctx.graph
        # the (masked) gradient in a buffer whose rows start on a 128-byte line: it is gathered and fed to the MFMA kernel
        line = 128 // g.element_size()
        pad = line if g.shape[1] < line else (16 // g.element_size())
        masked = ctx.relu and not ctx.grad_is_gated and not (ctx.token is not None and ctx.token.masked)   # (a loss that folds the ReLU in)
        if not masked and g.stride(1) == 1 and g.stride(0) % pad == 0 and g.data_ptr() % 16 == 0:
            gm = g                                             # already laid out that way (the loss kernel's gradient is)
        else:
            gm = ops.alloc_features(g.shape[0], g.shape[1], g.dtype, g.device, pad_to=pad)
            if masked:
                torch.ops.aten.threshold_backward.grad_input(g, out, 0, grad_input=gm)
            else:
                gm.copy_(g)
        # d/dz of reduce_A(z): A^T . g.  The mean's 1/deg rides as the edge values of the transposed structure (A^T edge (j <- i)
        # carries 1/deg(i), cached with the graph): 4 more bytes per edge in the narrow gather instead of a scaled copy of the
        # gradient -- a strided elementwise pass over [N, C] that cost 0.19 ms of the 21.7 ms step
        gt, _ = graph.transpose()
        if ctx.reduce == "mean" and gt.val is not None:        # a weighted adjacency: scale a copy of the gradient instead (a product of
            gp = ops.alloc_features(g.shape[0], g.shape[1], g.dtype, g.device, pad_to=pad)   # two per-edge arrays would be an nnz-sized pass)
            torch.mul(gm, (1.0 / graph.degrees().clamp(min=1).to(torch.float32)).unsqueeze(1).to(g.dtype), out=gp)
            gz = ops.spmm_raw(gt, gp, val=gt.val, reduce="sum")
        else:
            gz = ops.spmm_raw(gt, gm, val=graph.mean_scale_transposed() if ctx.reduce == "mean" else gt.val, reduce="sum")
        if ctx.needs_input_grad[1] and ctx.needs_input_grad[2]:
            # h^T . g and h^T . (A^T g): one launch that reads the wide operand h once (dense.grad_weight_shared_x)
            gws, gwn = dense.grad_weight_shared_x(h, gm, gz, out1=grad_slot_of(ctx.wparams[0]), out2=grad_slot_of(ctx.wparams[1]))
        else:
            gws = dense.grad_weight(h, gm, out=grad_slot_of(ctx.wparams[0])) if ctx.needs_input_grad[1] else None
            gwn = dense.grad_weight(h, gz, out=grad_slot_of(ctx.wparams[1])) if ctx.needs_input_grad[2] else None
        gh = None
        if ctx.needs_input_grad[0]:
            if dense._mfma_ok(gm, gz) and wsd.shape[0] <= 256 and (not ctx.gate_input or h.stride(1) == 1):
                # g.Ws^T + gz.Wn^T and the ReLU mask of the layer below: one MFMA launch, every operand read once
                gh = dense.transform_bf16(gm, wsd, gz, wnd, out_gate=h if ctx.gate_input else None, gate_bits=ctx.h_bits)
            else:
                gh = dense.mm2_nt(gm, wsd, gz, wnd, gate=h if ctx.gate_input else None)     # (fp32: the mask rides in the epilogue)
        return gh, gws, gwn, None, None, None, None, None, None, None


def can_fuse(layer, on_gpu):
    """The standard sageConv configuration the fused nodes implement."""
    from . import backend as F

    return bool(on_gpu and layer.aggr_hid_method == "sum" and not layer.neighborAgg.use_bias
                and layer.aggr_neighbor_method in ("mean", "sum") and layer.activation in (None, F.relu))


def sage_graph_layer(layer, graph, h, grad_is_gated=False, gate_input=False):
    """Full-graph sageConv (x_dst is x_src) through the fused nodes when the layer has the standard configuration;
    None otherwise (the caller falls back to forward_block).  grad_is_gated / gate_input: see the module docstring --
    only GraphSage.forward_graph, which owns the activations between its layers, sets them."""
    if not can_fuse(layer, h.is_cuda):
        return None
    relu = layer.activation is not None
    node = _SageGraphLayerTransformFirst if layer.transform_first(h) else _SageGraphLayer
    # a layer that will mask its incoming gradient itself tags its output: a consumer that takes the ReLU's backward over
    # (ops.cross_entropy(fold_relu=True)) says so through the token and the masking pass is skipped (ops.GateToken)
    token = ops.GateToken() if (relu and not grad_is_gated) else None
    box = [] if relu else None
    out = node.apply(h, layer.weight, layer.neighborAgg.weight, graph, layer.aggr_neighbor_method, relu,
                     bool(grad_is_gated and relu), bool(gate_input), token, box)
    if token is not None:
        out._dgll_gate_token = token
    if box:
        _tag_bits(out, box[0])
    return out
