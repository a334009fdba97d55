"""Operators of the aggregation hot path: thin autograd wrappers over the C ABI (include/dgll_hip.h).

Every function here takes CUDA (= HIP on torch-ROCm) tensors and launches hand-written gfx950 kernels on the
current torch stream.  There is no CPU implementation in this module and no fallback: a CPU tensor raises.
"""
import torch

from . import _lib
from .graph import CSRGraph

_DT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}
_REDUCE = {"sum": _lib.REDUCE_SUM, "mean": _lib.REDUCE_MEAN}


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("dgll_amd.ops run on the GPU only (got a %s tensor); there is no CPU fallback" % t.device)


def _dtype_code(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError("dgll_amd kernels take float32 or bfloat16 matrices, got %s" % t.dtype)


def _row_major(x):
    """2-D tensor with unit column stride (any row stride >= ncols is passed through as the leading dimension)."""
    if x.dim() != 2:
        raise ValueError("expected a 2-D matrix")
    if x.stride(1) != 1 or x.stride(0) < x.shape[1]:
        x = x.contiguous()
    return x


def alloc_features(n_rows, feat, dtype, device, pad_to=8):
    """[n_rows, feat] view of a buffer whose leading dimension is padded to 16 bytes, so any feature width
    takes the vectorised (global_load_dwordx4) path of the kernels."""
    ld = (feat + pad_to - 1) // pad_to * pad_to
    buf = torch.empty((n_rows, ld), dtype=dtype, device=device)
    return buf[:, :feat] if ld != feat else buf


class LaunchTimer:
    """HIP-event timing of individual kernel launches on the stream they are issued on (bench.py's roofline leg).
    Usage: `with LaunchTimer() as t: ...steps...` then t.summary() -> {tag: (count, avg_ms)}."""
    active = None

    def __init__(self):
        self.records = []

    def __enter__(self):
        LaunchTimer.active = self
        return self

    def __exit__(self, *exc):
        LaunchTimer.active = None

    def start(self, tag, device):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(torch.cuda.current_stream(device))
        self.records.append((tag, a, b))
        return b

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for tag, a, b in self.records:
            n, tot = out.get(tag, (0, 0.0))
            out[tag] = (n + 1, tot + a.elapsed_time(b))
        return {k: (n, tot / n) for k, (n, tot) in out.items()}


def spmm_raw(graph, x, val=None, reduce="sum", bias=None, relu=False, out_dtype=None, out=None, row_scale=None,
             accumulate=False, gate=None):
    """Y = epilogue(reduce_j A[i,j] X[j,:]) with no autograd.  `val` overrides graph.val (None = unweighted
    unless graph.val is set).  accumulate: add the rows already in `out` first; row_scale: fp32[n_rows] replacing the
    reduce's own scale (both used by the partitioned path, dgll_hip_spmm_csr_ex); accumulate=2 ("add"): the increment
    form out += gate(scale . A.X), rows without edges untouched.  gate: [n_rows, feat] of the output
    dtype -- outputs are zeroed where gate <= 0 (dgll_hip_spmm_csr_gated: the ReLU backward of the layer below)."""
    _require_cuda(x, graph.rowptr)
    x = _row_major(x)
    if x.shape[0] != graph.n_cols:
        raise ValueError("X has %d rows but the adjacency gathers from %d" % (x.shape[0], graph.n_cols))
    feat = x.shape[1]
    val = graph.val if val is None else val
    if val is not None:
        if val.dtype != torch.float32 or val.numel() != graph.nnz:
            raise ValueError("edge values must be fp32 with one entry per nonzero")
        val = val.contiguous()
    out_dtype = x.dtype if out_dtype is None else out_dtype
    if out is None:
        # (Round 6, measured and dropped: a pitch of whole 128-byte lines for rows a little over one line -- the 100-column bf16
        # aggregate at 256 instead of 208 bytes: the MFMA transform that reads it 0.614 -> 0.618 ms, the weight gradient 0.440 -> 0.465.)
        out = alloc_features(graph.n_rows, feat, out_dtype, x.device, pad_to=8 if out_dtype == torch.bfloat16 else 4)
    epi = (_lib.EPI_BIAS if bias is not None else 0) | (_lib.EPI_RELU if relu else 0)
    if bias is not None:
        bias = bias.detach().to(torch.float32).contiguous()
    plan = graph.plan()
    ws_bytes = graph.workspace_bytes(feat)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None
    timer = LaunchTimer.active
    if timer is not None:      # what distinguishes the launch kinds of a step: width, dtype, weights, nnz, epilogue extras
        extra = "+".join(t for t, on in (("accumulate", bool(accumulate)), ("gate", gate is not None),
                                         ("row_scale", row_scale is not None)) if on)
        end = timer.start(("spmm", feat, str(x.dtype), val is not None, graph.nnz, extra), x.device)
    else:
        end = None
    with _lib.on_device(x.device):
        stream = _lib.raw_stream(x.device)
        if gate is not None and (gate.dtype != out.dtype or gate.shape != out.shape or gate.stride(1) != 1):
            raise ValueError("gate must match the output's shape and dtype, rows contiguous")
        code = _lib.lib.dgll_hip_spmm_csr_gated(
            stream, plan, graph.rowptr.data_ptr(), graph.col.data_ptr(), val.data_ptr() if val is not None else None,
            x.data_ptr(), x.stride(0), _dtype_code(x), out.data_ptr(), out.stride(0), _dtype_code(out),
            graph.n_rows, graph.n_cols, feat, _REDUCE[reduce], epi, bias.data_ptr() if bias is not None else None,
            ws.data_ptr() if ws is not None else None, ws_bytes,
            row_scale.data_ptr() if row_scale is not None else None, 2 if accumulate in (2, "add") else int(bool(accumulate)),
            gate.data_ptr() if gate is not None else None, gate.stride(0) if gate is not None else 0)
    if end is not None:
        end.record(torch.cuda.current_stream(x.device))
    _lib.check(code, "dgll_hip_spmm_csr")
    return out


class _Spmm(torch.autograd.Function):
    """Y = A.X (+bias, ReLU) -- F.spmm of gcnconv.py:31 / SpecialSpmmFunction of gatconv.py:60-81.
    backward: grad_X = A^T.g through the cached transposed CSR (gatconv.py:80), grad_val = SDDMM(g, X)
    (gatconv.py:76-78), grad_bias = column sums."""

    @staticmethod
    def forward(ctx, x, val, bias, graph, reduce, relu, out_ref=None):
        # out_ref: a one-element list holding the destination (rows of a caller's buffer): a list, so that autograd does not take
        # the buffer for an input of the node
        y = spmm_raw(graph, x, val=val, reduce=reduce, bias=bias, relu=relu, out=out_ref[0] if out_ref else None)
        ctx.graph, ctx.reduce, ctx.relu = graph, reduce, relu
        ctx.x_dest = getattr(x, "_dgll_grad_dest", None)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x if (val is not None and val.requires_grad) else None, val, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, val, y = ctx.saved_tensors
        graph = ctx.graph
        if ctx.relu:
            g = torch.ops.aten.threshold_backward(g.contiguous(), y, 0)
        g = _row_major(g)
        grad_x = grad_val = grad_bias = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            from .dense import column_sum

            grad_bias = column_sum(g)
        if ctx.needs_input_grad[0] and graph.identity_cols and graph.n_cols == graph.nnz and g.is_cuda and val is None \
                and graph.val is None and g.dtype in (torch.float32, torch.bfloat16):
            # sampled block (col == arange), unweighted: source row e receives exactly scale . g[row(e)] -- ONE launch
            # (dgll_hip_expand_rows: the degree / reciprocal / scale / searchsorted / gather chain of tensor ops was nine launches per
            # block and batch, ~0.3 ms of a 1.7 ms sampled step); the unused tail of a block on static shapes comes out as zeros
            # written (or added) straight into the rows of the layer input's gradient buffer when the input is a row range of one
            # (ops.row_slices): no copy / add pass afterwards
            got = grad_dest(ctx.x_dest, (graph.n_cols, g.shape[1]), g.dtype, g.device)
            if got is not None:
                grad_x, acc = got
            else:
                grad_x, acc = alloc_features(graph.n_cols, g.shape[1], g.dtype, g.device, pad_to=16 // g.element_size()), False
            with _lib.on_device(g.device):
                _lib.check(_lib.lib.dgll_hip_expand_rows(_lib.raw_stream(g.device), graph.rowptr.data_ptr(), graph.n_rows, g.data_ptr(), g.stride(0),
                                                         grad_x.data_ptr(), grad_x.stride(0), graph.n_cols, g.shape[1], _dtype_code(g),
                                                         1 if ctx.reduce == "mean" else 0, 1 if acc else 0), "dgll_hip_expand_rows")
        elif ctx.needs_input_grad[0] and graph.identity_cols and graph.n_cols == graph.nnz:
            # the same through tensor ops (host tensors, weighted blocks): one gather,
            # no transposed CSR to sort together and no launch plan for it, both of which would be rebuilt every batch
            scale = (1.0 / graph.degrees().clamp(min=1).to(torch.float32)).unsqueeze(1).to(g.dtype) if ctx.reduce == "mean" else None
            if getattr(graph, "padded", False):
                # a block on static shapes (graphs.PaddedBlock): source rows past the batch's edges belong to no destination -- its
                # row_index() sends them to row n_rows, an extra all-zero row appended to the (scaled) gradient: exact zeros, no mask pass
                gs = torch.empty((g.shape[0] + 1, g.shape[1]), dtype=g.dtype, device=g.device)
                gs[g.shape[0]:].zero_()
                if scale is not None:
                    torch.mul(g, scale, out=gs[:g.shape[0]])
                else:
                    gs[:g.shape[0]].copy_(g)
            else:
                gs = g if scale is None else g * scale
            grad_x = gs.index_select(0, graph.row_index())
            w = graph.val if val is None else val.detach()
            if w is not None:
                grad_x = grad_x * w.unsqueeze(1).to(grad_x.dtype)
        elif ctx.needs_input_grad[0]:
            gt, perm = graph.transpose()
            tval = gt.val if val is None else val.detach()[perm]  # gt.val is the cached permuted graph.val
            gin = g
            if (g.stride(0) * g.element_size()) % 16 or g.data_ptr() % 16:
                # rows not 16-byte aligned (e.g. a 47-wide gradient): one copy into padded rows keeps the gather on
                # the vectorised kernel, and the mean's 1/deg(i) is folded into that copy instead of per-edge weights
                line = 128 // g.element_size()      # narrow rows: one 128-byte line each
                gin = alloc_features(g.shape[0], g.shape[1], g.dtype, g.device,
                                     pad_to=line if g.shape[1] < line else 16 // g.element_size())
                if ctx.reduce == "mean":
                    torch.mul(g, (1.0 / graph.degrees().clamp(min=1).to(torch.float32)).unsqueeze(1).to(g.dtype), out=gin)
                else:
                    gin.copy_(g)
            elif ctx.reduce == "mean":
                scale = graph.mean_scale_transposed()  # A^T edge (j <- i) carries 1/deg(i)
                tval = scale if tval is None else tval * scale
            grad_x = spmm_raw(gt, gin, val=tval, reduce="sum")
        if val is not None and ctx.needs_input_grad[1]:
            from .ops_edge import sddmm_raw

            grad_val = sddmm_raw(graph, g, x)
            if ctx.reduce == "mean":
                grad_val = grad_val / graph.degrees().clamp(min=1).to(torch.float32)[graph.row_index()]
        return grad_x, grad_val, grad_bias, None, None, None, None


def spmm(graph, x, val=None, reduce="sum", bias=None, relu=False, out=None):
    """Differentiable CSR SpMM on the GPU.  `graph` is a CSRGraph; `val` optional per-edge fp32 weights
    (defaults to graph.val); `reduce` 'sum' or 'mean'; optional fused bias / ReLU epilogue.  out: [n_rows, feat] rows of a
    caller's buffer to write (several reductions stacked in one buffer feed the next transform without a concatenation)."""
    if not isinstance(graph, CSRGraph):
        raise TypeError("spmm expects a CSRGraph (use dgll_amd.graph.as_csr_graph for torch sparse tensors)")
    if val is None and graph.val is not None and graph.val.requires_grad:
        val = graph.val
    return _Spmm.apply(x, val, bias, graph, reduce, relu, [out] if out is not None else None)


class _GradArena:
    """The gradient buffer of a matrix whose row ranges feed several consumers (row_slices): the consumers' backward passes may
    write their result straight into their range of it instead of returning a tensor that is copied / added in afterwards
    (five strided copies and an add per sampled step).  claim(k) -> (rows [a, b) of the buffer, accumulate) or None:
    accumulate = False when nothing has been written to the range yet (the producer WRITES), True when all of it has been
    written (the producer ADDS); a range that is partly written is not handed out (the node's backward merges that gradient)."""

    def __init__(self, shape, bounds):
        self.shape, self.bounds = tuple(shape), list(bounds)
        self.out = None
        self.written = []                  # disjoint, sorted [a, b)
        self.claimed = [False] * len(self.bounds)

    def _covered(self, a, b):
        pos, any_overlap = a, False
        for ca, cb in self.written:
            if cb <= a or ca >= b:
                continue
            any_overlap = True
            if ca > pos:
                return "partial"
            pos = max(pos, cb)
        if not any_overlap:
            return "none"
        return "all" if pos >= b else "partial"

    def _mark(self, a, b):
        self.written.append((a, b))
        self.written.sort()
        merged = [self.written[0]]
        for ca, cb in self.written[1:]:
            if ca <= merged[-1][1]:
                merged[-1] = (merged[-1][0], max(merged[-1][1], cb))
            else:
                merged.append((ca, cb))
        self.written = merged

    def claim(self, k, dtype, device, allow_accumulate=True):
        a, b = self.bounds[k]
        if b <= a or self.claimed[k]:
            return None
        state = self._covered(a, b)
        if state == "partial" or (state == "all" and not allow_accumulate):
            return None
        if self.out is None:
            self.out = alloc_features(self.shape[0], self.shape[1], dtype, device, pad_to=16 // torch.empty((), dtype=dtype).element_size())
        elif self.out.dtype != dtype:
            return None
        self.claimed[k] = True
        if state == "none":
            self._mark(a, b)
        return self.out[a:b], state == "all"


class _RowSlices(torch.autograd.Function):
    """Several row ranges [a, b) of one matrix, possibly overlapping, as one autograd node."""

    @staticmethod
    def forward(ctx, x, bounds, arena):
        ctx.bounds, ctx.shape = bounds, x.shape
        ctx.arena = arena
        return tuple(x[a:b] for a, b in bounds)

    @staticmethod
    def backward(ctx, *grads):
        # one buffer for the matrix's gradient.  A range whose producer CLAIMED its rows of the buffer (ctx.arena: the block's
        # expand launch, the transform's input-gradient product) is in place already; any other range's gradient is COPIED where
        # nothing has been written yet and ADDED where an earlier range overlaps; rows no range covers are zero-filled.  (Plain
        # slices: a zero-filled full-size tensor per range plus pairwise full-size additions.)
        arena = ctx.arena
        ref = next(g for g in grads if g is not None)
        out = arena.out if arena.out is not None else torch.empty(ctx.shape, dtype=ref.dtype, device=ref.device)
        covered = list(arena.written)                  # disjoint, sorted [a, b) ranges already written
        for k, ((a, b), g) in enumerate(zip(ctx.bounds, grads)):
            if g is None or b <= a:
                continue
            if arena.claimed[k]:
                if g.data_ptr() != out[a:b].data_ptr():
                    # (the view has a second consumer and autograd summed two gradients into a new tensor: the rows written in
                    # place hold only a part of it and cannot be told apart any more)
                    raise RuntimeError("row_slices: a range whose gradient was written in place arrived as another tensor "
                                       "(a view returned by row_slices must feed exactly one differentiable consumer)")
                continue                               # written (or added) in place by its producer
            pos = a
            for ca, cb in list(covered):
                if cb <= pos or ca >= b:
                    continue
                if ca > pos:                           # a gap before this covered piece: first write
                    out[pos:ca].copy_(g[pos - a:ca - a])
                lo, hi = max(ca, pos), min(cb, b)
                out[lo:hi].add_(g[lo - a:hi - a])
                pos = hi
            if pos < b:
                out[pos:b].copy_(g[pos - a:b - a])
            covered.append((a, b))
            covered.sort()
            merged = [covered[0]]
            for ca, cb in covered[1:]:
                if ca <= merged[-1][1]:
                    merged[-1] = (merged[-1][0], max(merged[-1][1], cb))
                else:
                    merged.append((ca, cb))
            covered = merged
        pos = 0
        for ca, cb in covered + [(ctx.shape[0], ctx.shape[0])]:
            if ca > pos:
                out[pos:ca].zero_()
            pos = max(pos, cb)
        ctx.arena = None
        return out, None, None


def row_slices(x, bounds):
    """[x[a:b] for (a, b) in bounds] through one autograd node (GraphSage._forward_sampled_batched: the stacked hops of a layer's
    input are read as overlapping row ranges)."""
    bounds = [(int(a), int(b)) for a, b in bounds]
    if not x.requires_grad:
        return [x[a:b] for a, b in bounds]
    arena = _GradArena(x.shape, bounds)
    outs = list(_RowSlices.apply(x, bounds, arena))
    if x.is_cuda:
        for k, o in enumerate(outs):           # a consumer whose backward can write its gradient in place looks for this (grad_dest)
            o._dgll_grad_dest = (arena, k)
    return outs


def grad_dest(dest, shape, dtype, device, allow_accumulate=True):
    """(rows of a row_slices gradient buffer to write this gradient into, accumulate) for a producer's backward pass, or None."""
    if dest is None:
        return None
    arena, k = dest
    a, b = arena.bounds[k]
    if (b - a, arena.shape[1]) != tuple(shape):
        return None
    return arena.claim(k, dtype, device, allow_accumulate=allow_accumulate)


# ------------------------------------------------------------------------------------------------ loss
class GateToken:
    """Handshake between a layer that ends in a ReLU and a consumer willing to take that ReLU's backward over (cross_entropy's
    fold_relu): the layer tags its output with a token; a consumer that masks the gradient it returns sets `masked`, and the
    layer's backward then skips its own masking pass.  No tag, or nobody sets it: the layer masks as always."""
    __slots__ = ("masked",)

    def __init__(self):
        self.masked = False


def _xent_launch(z, target, soft, row_loss, grad, scale, mask_nonpositive=False, own_padding=False):
    n, c = z.shape
    with _lib.on_device(z.device):
        stream = _lib.raw_stream(z.device)
        rl = row_loss.data_ptr() if row_loss is not None else None
        gp, gld = (grad.data_ptr(), grad.stride(0)) if grad is not None else (None, 0)
        sp = scale.data_ptr() if scale is not None else None
        if mask_nonpositive or own_padding:      # own_padding: `grad` is a fresh padded allocation whose padding may be written (zeros)
            code = _lib.lib.dgll_hip_softmax_xent_ex(stream, z.data_ptr(), z.stride(0), _dtype_code(z), None if soft else target.data_ptr(),
                                                     target.data_ptr() if soft else None, target.stride(0) if soft else 0, rl, gp, gld,
                                                     sp, n, c, (1 if mask_nonpositive else 0) | (2 if own_padding else 0))
        elif soft:
            code = _lib.lib.dgll_hip_softmax_xent_soft(stream, z.data_ptr(), z.stride(0), _dtype_code(z), target.data_ptr(),
                                                       target.stride(0), rl, gp, gld, sp, n, c)
        else:
            code = _lib.lib.dgll_hip_softmax_xent(stream, z.data_ptr(), z.stride(0), _dtype_code(z), target.data_ptr(),
                                                  rl, gp, gld, sp, n, c)
    _lib.check(code, "dgll_hip_softmax_xent")


XENT_ONE_LAUNCH_ROWS = 1 << 16     # up to here the loss of a batch is finished by one workgroup (larger: the two-stage tree below)


class _CrossEntropy(torch.autograd.Function):
    """Softmax cross-entropy, one kernel per direction (dgll_hip_softmax_xent / _soft).  The per-row losses are summed in
    two stages (a [n] -> [~sqrt n] -> scalar tree of small reductions) rather than by one large single-output reduction:
    the latter relies on a memset node when captured in a HIP graph, which replays unreliably on this stack
    (dgll_amd/graphs.py)."""

    @staticmethod
    def forward(ctx, logits, target, reduction, token=None):
        ctx.token = token
        _require_cuda(logits, target)
        if logits.dim() != 2:
            raise ValueError("cross_entropy expects logits [N, C]")
        if logits.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError("cross_entropy: fp32 or bf16 logits")
        soft = target.is_floating_point()
        if soft:
            if target.shape != logits.shape:
                raise ValueError("probability targets must have the logits' shape [N, C]")
            target = target.to(torch.float32)
            target = target if target.stride(1) == 1 else target.contiguous()
        else:
            if target.shape != (logits.shape[0],):
                raise ValueError("cross_entropy expects logits [N, C] and labels [N]")
            target = target.to(torch.int64).contiguous()
        z = logits if logits.stride(1) == 1 else logits.contiguous()
        n, c = z.shape
        if not soft and reduction in ("mean", "sum") and 0 < n <= XENT_ONE_LAUNCH_ROWS:
            # a mini-batch: {total, count, mean, 1 / count} by ONE native launch (dgll_hip_xent_reduce) instead of nine tensor ops
            row_loss = torch.empty(n, dtype=torch.float32, device=z.device)
            _xent_launch(z, target, soft, row_loss, None, None)
            res = torch.empty(4, dtype=torch.float32, device=z.device)
            with _lib.on_device(z.device):
                _lib.check(_lib.lib.dgll_hip_xent_reduce(_lib.raw_stream(z.device), row_loss.data_ptr(), target.data_ptr(), n, c, res.data_ptr()),
                           "dgll_hip_xent_reduce")
            ctx.reduction, ctx.soft = reduction, soft
            ctx.inv_count = res[3] if reduction == "mean" else None
            ctx.save_for_backward(z, target, None)
            return res[2] if reduction == "mean" else res[0]
        ctx.inv_count = None
        store = torch.empty(-(-n // 256) * 256, dtype=torch.float32, device=z.device)   # padded for the two-stage sum
        store[n:].zero_()
        row_loss = store[:n]
        _xent_launch(z, target, soft, row_loss, None, None)
        ctx.reduction, ctx.soft = reduction, soft
        count = None
        if reduction == "mean":          # torch semantics: mean over the targets that are not ignored (soft targets: over N)
            if soft:
                count = torch.full((), float(n), device=z.device)
            else:
                # the same two-stage tree as the loss total (a single-output reduction over [n] replays as 0 inside a
                # captured HIP graph on this stack, which would turn the loss and every gradient into inf / NaN)
                valid = torch.zeros(store.numel(), dtype=torch.float32, device=z.device)
                valid[:n] = ((target >= 0) & (target < c))
                count = valid.view(-1, 256).sum(1).sum() if n > 4096 else valid[:n].sum()
        ctx.save_for_backward(z, target, count)
        if reduction == "none":
            return row_loss
        total = store.view(-1, 256).sum(1).sum() if n > 4096 else row_loss.sum()
        return total / count if reduction == "mean" else total

    @staticmethod
    def backward(ctx, g):
        z, target, count = ctx.saved_tensors
        per_row = ctx.reduction == "none"
        if per_row:
            scale = None
        elif ctx.inv_count is not None:
            scale = (g.float() * ctx.inv_count).reshape(-1)
        else:
            scale = (g.float() / count if count is not None else g.float()).reshape(-1)
        # rows of the gradient start on 16-byte (narrow bf16 logits: 128-byte) boundaries: the layer below gathers it and feeds
        # it to the MFMA kernels as it is, without a re-layout pass over [N, C]
        line = 128 // z.element_size()
        grad = alloc_features(z.shape[0], z.shape[1], z.dtype, z.device, pad_to=line if z.shape[1] < line else 16 // z.element_size())
        # fold_relu: the logits are a ReLU's output and this pass returns the gradient of the PRE-activation (zero where z <= 0);
        # the producing layer is told through its token and skips its own masking pass over [N, C]
        _xent_launch(z, target, ctx.soft, None, grad, scale, mask_nonpositive=ctx.token is not None, own_padding=True)
        if ctx.token is not None:
            ctx.token.masked = True
        if per_row:
            grad = grad * g.to(grad.dtype).unsqueeze(1)
        else:
            c = z.shape[1]
            c8 = -(-c // 8) * 8
            if (z.dtype == torch.bfloat16 and not ctx.soft and c <= 512 and z.data_ptr() % 16 == 0 and z.stride(0) % 8 == 0
                    and z.stride(0) >= c8 and grad.stride(0) >= c8):
                # loss.hip's vector form (exactly these conditions, xent_impl) stores whole 16-byte vectors: the columns behind the C
                # classes up to the next multiple of 8 are ZERO.  A consumer that needs the gradient that much wider (a layer computed on
                # padded rows: gatconv._UnpadOneHead) may re-view this buffer instead of copying it; (columns, storage pointer,
                # version) -- valid only for this very storage and version, as fused_layers._tag_bits
                grad._dgll_zero_padding = (c8, grad.data_ptr(), grad._version)
        return grad, None, None, None


def cross_entropy(logits, labels, reduction="mean", fold_relu=False):
    """F.cross_entropy(logits, target) for GPU logits [N, C] (fp32 or bf16; math in fp32).  `target`: int64 class indices
    [N] (entries outside [0, C), e.g. -100, are ignored) or a float [N, C] matrix of probabilities / multi-hot labels --
    the PPI loop's case (Evaluation/PPI/train_gcn.py:27,45 hands nn.CrossEntropyLoss the float label matrix).
    fold_relu: the caller asserts that `logits` -- the output of a layer that ends in a ReLU, as the reference's GraphSage does
    (sageconv.py:83) -- has NO other differentiable consumer.  When that layer tagged its output (GateToken), the gradient pass of
    the loss also applies the ReLU's mask and the layer skips its own pass over [N, C]; same gradients (masking is idempotent), one
    elementwise launch less.  Untagged logits: ignored."""
    if reduction not in ("mean", "sum", "none"):
        raise ValueError("reduction must be 'mean', 'sum' or 'none'")
    token = getattr(logits, "_dgll_gate_token", None) if fold_relu else None
    return _CrossEntropy.apply(logits, labels, reduction, token)


from .ops_edge import gat_aggregate, gat_layer, head_width_padded, sddmm_raw, segment_max  # noqa: E402,F401
