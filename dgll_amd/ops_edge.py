"""Per-edge operators of the GAT / SpecialSpmm path (C ABI: dgll_hip_sddmm_csr, dgll_hip_gat_fwd/bwd,
dgll_hip_segment_max).  GPU only; no fallback."""
import os

import torch

from . import _lib
from .graph import CSRGraph
from .ops import _dtype_code, _require_cuda


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def _epv(dtype):
    return 8 if dtype == torch.bfloat16 else 4


def _vec_ready(x):
    """16-byte aligned rows: unit column stride, leading dimension a multiple of one vector."""
    esz = x.element_size()
    return (x.dim() == 2 and x.stride(1) == 1 and x.stride(0) >= x.shape[1] and (x.stride(0) * esz) % 16 == 0
            and x.data_ptr() % 16 == 0 and x.stride(0) >= -(-x.shape[1] // (16 // esz)) * (16 // esz))


def _padded_copy(x, width=None):
    """Copy of x whose leading dimension (and optionally logical width, zero filled) is vector aligned."""
    epv = _epv(x.dtype)
    width = x.shape[1] if width is None else width
    ld = -(-width // epv) * epv
    buf = torch.zeros((x.shape[0], ld), dtype=x.dtype, device=x.device)
    buf[:, :x.shape[1]] = x
    return buf[:, :width] if ld != width else buf


def _ready(x):
    return x if _vec_ready(x) else _padded_copy(x)


def _empty_padded(n, width, dtype, device):
    epv = _epv(dtype)
    ld = -(-width // epv) * epv
    buf = torch.empty((n, ld), dtype=dtype, device=device)
    return buf[:, :width] if ld != width else buf


# ------------------------------------------------------------------------------------------------ SDDMM
def sddmm_raw(graph, g, b):
    """edge_out[k] = <g[row(k)], b[col[k]]> -- SpecialSpmmFunction.backward's grad_values (gatconv.py:76-78)."""
    _require_cuda(g, b, graph.rowptr)
    if g.dtype != b.dtype:
        b = b.to(g.dtype)
    feat = g.shape[1]
    tile = 64 * _epv(g.dtype)
    out = torch.empty(graph.nnz, dtype=torch.float32, device=g.device)
    total = None
    for c0 in range(0, feat, tile):  # > 64 vectors per row: column blocks, summed
        gs, bs = _ready(g[:, c0:c0 + tile]), _ready(b[:, c0:c0 + tile])
        with torch.cuda.device(g.device):
            code = _lib.lib.dgll_hip_sddmm_csr(_stream(g.device), graph.rowptr.data_ptr(), graph.col.data_ptr(),
                                               gs.data_ptr(), gs.stride(0), bs.data_ptr(), bs.stride(0), _dtype_code(gs),
                                               out.data_ptr(), graph.n_rows, gs.shape[1])
        _lib.check(code, "dgll_hip_sddmm_csr")
        if feat > tile:
            total = out.clone() if total is None else total + out
    return out if total is None else total


# ------------------------------------------------------------------------------------------------ segment max
class _SegmentMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, graph):
        _require_cuda(x, graph.rowptr)
        if x.shape[0] != graph.n_cols:      # the backward walks the transposed structure: one row pointer per source row
            raise ValueError("X has %d rows but the adjacency gathers from %d" % (x.shape[0], graph.n_cols))
        xs = _ready(x)
        feat = x.shape[1]
        y = _empty_padded(graph.n_rows, feat, x.dtype, x.device)
        arg = torch.empty((graph.n_rows, y.stride(0)), dtype=torch.int32, device=x.device)
        with torch.cuda.device(x.device):
            code = _lib.lib.dgll_hip_segment_max(_stream(x.device), graph.rowptr.data_ptr(), graph.col.data_ptr(),
                                                 xs.data_ptr(), xs.stride(0), y.data_ptr(), arg.data_ptr(), y.stride(0),
                                                 _dtype_code(xs), graph.n_rows, feat)
        _lib.check(code, "dgll_hip_segment_max")
        ctx.save_for_backward(arg)
        ctx.n_src, ctx.graph, ctx.feat = x.shape[0], graph, feat
        return y

    @staticmethod
    def backward(ctx, g):
        """The gradient goes to the arg-max row only: a gather over the transposed structure (dgll_hip_segment_max_bwd),
        one wavefront per source row, no atomics.  Sampled blocks (col == arange: every source row has exactly one
        in-edge) need no sort: their transposed structure is rowptr = arange, col = the row of each edge."""
        (arg,) = ctx.saved_tensors
        graph, feat = ctx.graph, ctx.feat
        if graph.identity_cols and graph.n_cols == graph.nnz:
            t_rowptr = torch.arange(ctx.n_src + 1, dtype=torch.int64, device=g.device)
            t_col = graph.row_index().to(torch.int32)
        else:
            gt, _ = graph.transpose()
            t_rowptr, t_col = gt.rowptr, gt.col
        gs = _ready(g.to(arg.device))
        grad = _empty_padded(ctx.n_src, feat, g.dtype, g.device)
        with torch.cuda.device(g.device):
            code = _lib.lib.dgll_hip_segment_max_bwd(_stream(g.device), t_rowptr.data_ptr(), t_col.data_ptr(), gs.data_ptr(),
                                                     gs.stride(0), arg.data_ptr(), arg.stride(0), grad.data_ptr(),
                                                     grad.stride(0), _dtype_code(gs), ctx.n_src, feat)
        _lib.check(code, "dgll_hip_segment_max_bwd")
        return grad, None


def segment_max(graph, x):
    """max over each row's neighbours -- NeighborAggregator 'max' (sageconv.py:37-38); empty rows give 0."""
    return _SegmentMax.apply(x, graph)


# ------------------------------------------------------------------------------------------------ fused GAT
def head_width_padded(fo, dtype, pow2=True):
    """Per-head column count the kernels need: a whole number of 16-byte vectors -- a power of two of them for the
    first-generation kernels (max-subtracted softmax, attention dropout), any number for sparseGatConv's form."""
    epv = _epv(dtype)
    vecs = -(-fo // epv)
    if not pow2:
        return vecs * epv
    p = 1
    while p < vecs:
        p <<= 1
    return p * epv


def _row_slot(x, byte_off, n_floats):
    """fp32 [rows, n_floats] alias of the bytes [byte_off, byte_off + 4 n_floats) of every row of `x`'s storage -- a slot in
    the padding between a row's last column and the next row (the caller owns that padding)."""
    esz = x.element_size()
    base, stride = x.storage_offset() * esz + byte_off, x.stride(0) * esz
    if base % 4 or stride % 4 or byte_off + 4 * n_floats > stride:
        raise ValueError("no room for a %d-float slot at byte %d of %d-byte rows" % (n_floats, byte_off, stride))
    return torch.empty(0, dtype=torch.float32, device=x.device).set_(x.untyped_storage(), base // 4, (x.shape[0], n_floats),
                                                                    (stride // 4, 1))


def _empty_like_rows(n, like):
    """[n, like.shape[1]] with `like`'s row stride (the same padding behind every row)."""
    buf = torch.empty((n, like.stride(0)), dtype=like.dtype, device=like.device)
    return buf[:, :like.shape[1]] if like.stride(0) != like.shape[1] else buf


ROW_SCORES = os.environ.get("DGLL_GAT_ROW_SCORES", "1") != "0"     # 0: the forward always gathers T (A/B, tests)
ROW_SCORES_BWD = os.environ.get("DGLL_GAT_ROW_SCORES_BWD", "1") != "0"   # the rows pass of the backward in the same form (6.07 -> 5.68 ms)


def _gat_strided_forward(h, s, t, graph, heads, fo, alpha, apply_elu, pack_scores, attn2=None):
    """Forward gather pass on dgll_hip_gat_fwd_strided.  Returns (h with aligned rows, s, t, out, rowsum, packed).
    attn2: fp32 [heads * fo], a2 of every head laid out like a row of h (t = h . a2 per head): when the scores cannot ride in the rows'
    padding the pass forms t_j from the row it gathers anyway instead of fetching T[j] (dgll_hip_gat_fwd_rowscore; the reference builds
    its logit from the gathered rows too, gatconv.py:122-125)."""
    _require_cuda(h, s, t, graph.rowptr)
    dev = h.device
    h = _ready(h)
    esz, width = h.element_size(), heads * fo
    s = s.to(torch.float32).contiguous()
    t = t.to(torch.float32).contiguous()
    packed = bool(pack_scores) and (h.stride(0) - width) * esz >= 4 * heads
    if packed:
        t_gather = _row_slot(h, width * esz, heads)
        t_gather.copy_(t)
    else:
        t_gather = t
    out = _empty_like_rows(graph.n_rows, h)
    rowsum = torch.empty((graph.n_rows, heads), dtype=torch.float32, device=dev)
    plan = graph.plan()
    ws_bytes = int(_lib.lib.dgll_hip_gat_workspace_bytes(plan, heads, fo))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
    timer = _launch_timer()
    # (the row-score kernel addresses h by 32-bit byte offsets: at most 2^24 rows and 4 GB)
    row_scores = attn2 is not None and not packed and ROW_SCORES and h.shape[0] <= (1 << 24) and h.stride(0) * esz < (1 << 24) \
        and h.shape[0] * h.stride(0) * esz <= 0xFFFFFFFF
    end = timer.start(("gat", "fwd", heads, fo, str(h.dtype), graph.nnz, "packed" if packed else ("rowscore" if row_scores else "")), dev) if timer else None
    with torch.cuda.device(dev):
        if row_scores:
            a2 = attn2.detach().to(torch.float32).contiguous()
            code = _lib.lib.dgll_hip_gat_fwd_rowscore(
                _stream(dev), plan, graph.rowptr.data_ptr(), graph.col.data_ptr(), h.data_ptr(), h.stride(0), s.data_ptr(),
                a2.data_ptr(), out.data_ptr(), out.stride(0), _dtype_code(h), rowsum.data_ptr(), graph.n_rows, int(h.shape[0]), heads, fo,
                float(alpha), int(apply_elu), ws.data_ptr() if ws is not None else None, ws_bytes, 0, 0)
        else:
            code = _lib.lib.dgll_hip_gat_fwd_strided(
                _stream(dev), plan, graph.rowptr.data_ptr(), graph.col.data_ptr(), h.data_ptr(), h.stride(0), s.data_ptr(),
                t_gather.data_ptr(), t_gather.stride(0), out.data_ptr(), out.stride(0), _dtype_code(h), rowsum.data_ptr(),
                graph.n_rows, heads, fo, float(alpha), int(apply_elu), ws.data_ptr() if ws is not None else None, ws_bytes)
    if end is not None:
        end.record(torch.cuda.current_stream(dev))
    _lib.check(code, "dgll_hip_gat_fwd_rowscore" if row_scores else "dgll_hip_gat_fwd_strided")
    return h, s, t, out, rowsum, packed


def _gat_strided_backward(g, h, s, t, out, rowsum, graph, heads, fo, alpha, apply_elu, packed, attn=None, rows_first=False):
    """Both backward gather passes (dgll_hip_gat_bwd_rows_strided, _cols_strided).  attn = (a1, a2) fp32 [heads * fo]: the
    scores are S = H.a1, T = H.a2 per head and their contribution to grad_h is added by the second pass's epilogue.
    Returns (grad_h, grad_s, grad_t)."""
    dev, esz, width = h.device, h.element_size(), heads * fo
    g = _ready(g.to(h.dtype))
    gt, _ = graph.transpose()
    dn = _empty_like_rows(graph.n_rows, h)
    t_gather = _row_slot(h, width * esz, heads) if packed else t       # written by the forward; h's padding is ours
    if packed and (dn.stride(0) - width) * esz >= 8 * heads:
        sd = _row_slot(dn, width * esz, 2 * heads)
    else:
        sd = torch.empty((graph.n_rows, 2 * heads), dtype=torch.float32, device=dev)
    grad_h = _empty_like_rows(graph.n_cols, h)
    grad_s = torch.empty((graph.n_rows, heads), dtype=torch.float32, device=dev)
    grad_t = torch.empty((graph.n_cols, heads), dtype=torch.float32, device=dev)
    plan, t_plan = graph.plan(), gt.plan()
    ws_bytes = max(int(_lib.lib.dgll_hip_gat_workspace_bytes(plan, heads, fo)),
                   int(_lib.lib.dgll_hip_gat_workspace_bytes(t_plan, heads, fo)))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
    timer = _launch_timer()
    tag = (heads, fo, str(h.dtype), graph.nnz, "packed" if packed else "")
    wsp = ws.data_ptr() if ws is not None else None
    a1 = a2 = None
    gs_cols = grad_s
    if attn is not None:
        if graph.n_rows != graph.n_cols:
            # a rectangular adjacency whose rows ARE its first n_rows columns (a rank's merged adjacency, dist.py: own rows, then halo
            # rows): the transposed pass owns n_cols rows, grad_S exists for the first n_rows of them -- nobody uses a halo node's s here
            if not rows_first or graph.n_rows > graph.n_cols:
                raise ValueError("the score-gradient epilogue needs a square adjacency (grad_S of the rows the transposed pass owns)")
            gs_cols = torch.zeros((graph.n_cols, heads), dtype=torch.float32, device=dev)
        a1, a2 = (v.detach().to(torch.float32).contiguous() for v in attn)
    with torch.cuda.device(dev):
        st = _stream(dev)
        small = h.shape[0] <= (1 << 24) and h.stride(0) * esz < (1 << 24) and h.shape[0] * h.stride(0) * esz <= 0xFFFFFFFF
        rows_rowscore = a2 is not None and not packed and ROW_SCORES_BWD and small
        end = timer.start(("gat", "bwd_rows") + (tag[:4] + ("rowscore",) if rows_rowscore else tag), dev) if timer else None
        if rows_rowscore:      # t_j from the gathered rows, as in the forward
            code = _lib.lib.dgll_hip_gat_bwd_rows_rowscore(
                st, plan, graph.rowptr.data_ptr(), graph.col.data_ptr(), h.data_ptr(), h.stride(0), s.data_ptr(), a2.data_ptr(),
                out.data_ptr(), out.stride(0), g.data_ptr(), g.stride(0), _dtype_code(h), rowsum.data_ptr(), dn.data_ptr(), dn.stride(0),
                sd.data_ptr(), sd.stride(0), grad_s.data_ptr(), graph.n_rows, int(h.shape[0]), heads, fo, alpha, apply_elu, wsp, ws_bytes)
        else:
            code = _lib.lib.dgll_hip_gat_bwd_rows_strided(
                st, plan, graph.rowptr.data_ptr(), graph.col.data_ptr(), h.data_ptr(), h.stride(0), s.data_ptr(),
                t_gather.data_ptr(), t_gather.stride(0), out.data_ptr(), out.stride(0), g.data_ptr(), g.stride(0), _dtype_code(h),
                rowsum.data_ptr(), dn.data_ptr(), dn.stride(0), sd.data_ptr(), sd.stride(0), grad_s.data_ptr(), graph.n_rows,
                heads, fo, alpha, apply_elu, wsp, ws_bytes)
        if end is not None:
            end.record(torch.cuda.current_stream(dev))
        _lib.check(code, "dgll_hip_gat_bwd_rows_rowscore" if rows_rowscore else "dgll_hip_gat_bwd_rows_strided")
        if gs_cols is not grad_s:
            gs_cols[:graph.n_rows].copy_(grad_s)
        end = timer.start(("gat", "bwd_cols") + tag, dev) if timer else None
        code = _lib.lib.dgll_hip_gat_bwd_cols_strided(
            st, t_plan, gt.rowptr.data_ptr(), gt.col.data_ptr(), dn.data_ptr(), dn.stride(0), h.data_ptr(), h.stride(0),
            t.data_ptr(), sd.data_ptr(), sd.stride(0), grad_h.data_ptr(), grad_h.stride(0), grad_t.data_ptr(), _dtype_code(h),
            graph.n_cols, heads, fo, alpha, wsp, ws_bytes, a1.data_ptr() if a1 is not None else None,
            a2.data_ptr() if a2 is not None else None, gs_cols.data_ptr() if a1 is not None else None)
        if end is not None:
            end.record(torch.cuda.current_stream(dev))
        _lib.check(code, "dgll_hip_gat_bwd_cols_strided")
    return grad_h, grad_s, grad_t


class _GatAggregateStrided(torch.autograd.Function):
    """sparseGatConv's form (exp(-leakyrelu), no attention dropout) on the second-generation kernels
    (dgll_hip_gat_fwd_strided / _bwd_*_strided).  These passes are bound by cache-line fills per edge, so per-node scalars
    that are gathered per edge live next to what is gathered anyway: with `pack_scores` (the caller owns the padding behind
    h's rows) T sits in the padding of the feature rows -- a 47-class output row is 96 of 128 bytes -- and the backward keeps
    {s_i, dd_i} in the padding of its DN rows; without room they are compact / side-by-side arrays."""

    @staticmethod
    def forward(ctx, h, s, t, graph, heads, fo, alpha, apply_elu, pack_scores):
        h, s, t, out, rowsum, packed = _gat_strided_forward(h, s, t, graph, heads, fo, alpha, apply_elu, pack_scores)
        ctx.graph, ctx.cfg, ctx.packed = graph, (heads, fo, float(alpha), int(apply_elu)), packed
        ctx.save_for_backward(h, s, t, out, rowsum)
        return out

    @staticmethod
    def backward(ctx, g):
        h, s, t, out, rowsum = ctx.saved_tensors
        heads, fo, alpha, apply_elu = ctx.cfg
        grad_h, grad_s, grad_t = _gat_strided_backward(g, h, s, t, out, rowsum, ctx.graph, heads, fo, alpha, apply_elu, ctx.packed)
        return grad_h, grad_s, grad_t, None, None, None, None, None, None


class _GatLayerStrided(torch.autograd.Function):
    """The whole attention layer after the transform: scores S = H.a1, T = H.a2 per head (one skinny product
    H . blockdiag(a1 | a2), gatconv.py:122-125 without the [2 fo, E] edge matrix), then the aggregation above.  As ONE autograd
    node the backward needs no [n, 2 heads] x [2 heads, heads * fo] product and no add over [n, heads * fo] for the scores'
    contribution to grad_H: the transposed gather pass adds grad_S * a1 + grad_T * a2 in its epilogue."""

    @staticmethod
    def forward(ctx, h, A, graph, heads, fo, alpha, apply_elu, pack_scores):
        """h: one row per COLUMN of `graph`; a rectangular graph's rows are its first n_rows columns (dist.py's merged adjacency)."""
        from . import dense

        h = _ready(h)
        if h.shape[0] != graph.n_cols or graph.n_rows > graph.n_cols:
            raise ValueError("gat_layer: h needs one row per column of the adjacency, whose rows are its first columns")
        Ad = A.detach().to(h.dtype)
        st = (dense.transform_bf16(h, Ad.t(), out_dtype=torch.float32) if (dense._mfma_ok(h) and Ad.shape[1] <= 256)
              else dense.mm_nt(h, Ad.t()).float())
        # a2 of every head laid out like a row of h (A is block-diagonal: column heads + k holds a2 of head k in rows k fo .. (k + 1) fo)
        h, s, t, out, rowsum, packed = _gat_strided_forward(h, st[:graph.n_rows, :heads], st[:, heads:], graph, heads, fo, alpha, apply_elu,
                                                            pack_scores, attn2=Ad[:, heads:].float().sum(1))
        ctx.graph, ctx.cfg, ctx.packed = graph, (heads, fo, float(alpha), int(apply_elu)), packed
        ctx.save_for_backward(h, Ad, s, t, out, rowsum)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import dense

        h, Ad, s, t, out, rowsum = ctx.saved_tensors
        heads, fo, alpha, apply_elu = ctx.cfg
        # block-diagonal A: column k holds a1 of head k in rows k fo .. (k + 1) fo, column heads + k holds a2
        a1 = Ad[:, :heads].float().sum(1)
        a2 = Ad[:, heads:].float().sum(1)
        grad_h, grad_s, grad_t = _gat_strided_backward(g, h, s, t, out, rowsum, ctx.graph, heads, fo, alpha, apply_elu, ctx.packed,
                                                       attn=(a1, a2) if ctx.needs_input_grad[0] else None, rows_first=True)
        grad_A = None
        if ctx.needs_input_grad[1]:      # dA = H^T . [grad_S | grad_T], columns padded to a 16-byte row for the split-K kernel
            n = 2 * heads
            pad = -n % 8
            gst = torch.zeros((h.shape[0], n + pad), dtype=h.dtype, device=h.device)
            gst[:grad_s.shape[0], :heads] = grad_s
            gst[:, heads:n] = grad_t
            grad_A = dense.grad_weight(h, gst)[:, :n]
        return (grad_h if ctx.needs_input_grad[0] else None), grad_A, None, None, None, None, None, None


def _launch_timer():
    from .ops import LaunchTimer

    return LaunchTimer.active


class _GatAggregate(torch.autograd.Function):
    """out = act( sum_j w_ij scale_ij h_j / sum_j w_ij ) for all heads at once; see include/dgll_hip.h."""

    @staticmethod
    def forward(ctx, h, s, t, edge_scale, graph, heads, fo, alpha, apply_elu, mode):
        _require_cuda(h, s, t, graph.rowptr)
        dev = h.device
        h = _ready(h)
        s = s.to(torch.float32).contiguous()
        t = t.to(torch.float32).contiguous()
        out = _empty_padded(graph.n_rows, heads * fo, h.dtype, dev)
        rowsum = torch.empty((graph.n_rows, heads), dtype=torch.float32, device=dev)
        rowmax = torch.empty((graph.n_rows, heads), dtype=torch.float32, device=dev) if mode == 1 else None
        if edge_scale is not None:
            edge_scale = edge_scale.to(torch.float32).contiguous()
        plan = graph.plan()
        ws_bytes = int(_lib.lib.dgll_hip_gat_workspace_bytes(plan, heads, fo))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
        with torch.cuda.device(dev):
            code = _lib.lib.dgll_hip_gat_fwd(
                _stream(dev), plan, graph.rowptr.data_ptr(), graph.col.data_ptr(), h.data_ptr(), h.stride(0), s.data_ptr(),
                t.data_ptr(), edge_scale.data_ptr() if edge_scale is not None else None, out.data_ptr(), out.stride(0),
                _dtype_code(h), rowsum.data_ptr(), rowmax.data_ptr() if rowmax is not None else None, graph.n_rows, heads,
                fo, float(alpha), int(apply_elu), int(mode), ws.data_ptr() if ws is not None else None, ws_bytes)
        _lib.check(code, "dgll_hip_gat_fwd")
        ctx.graph, ctx.cfg = graph, (heads, fo, float(alpha), int(apply_elu), int(mode))
        ctx.save_for_backward(h, s, t, edge_scale, out, rowsum, rowmax)
        return out

    @staticmethod
    def backward(ctx, g):
        h, s, t, edge_scale, out, rowsum, rowmax = ctx.saved_tensors
        graph = ctx.graph
        heads, fo, alpha, apply_elu, mode = ctx.cfg
        dev = h.device
        g = _ready(g.to(h.dtype))
        gt, perm = graph.transpose()
        dn = _empty_padded(graph.n_rows, heads * fo, h.dtype, dev)
        dd = torch.empty((graph.n_rows, heads), dtype=torch.float32, device=dev)
        grad_h = _empty_padded(graph.n_cols, heads * fo, h.dtype, dev)
        grad_s = torch.empty((graph.n_rows, heads), dtype=torch.float32, device=dev)
        grad_t = torch.empty((graph.n_cols, heads), dtype=torch.float32, device=dev)
        plan, t_plan = graph.plan(), gt.plan()
        ws_bytes = max(int(_lib.lib.dgll_hip_gat_workspace_bytes(plan, heads, fo)),
                       int(_lib.lib.dgll_hip_gat_workspace_bytes(t_plan, heads, fo)))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
        with torch.cuda.device(dev):
            code = _lib.lib.dgll_hip_gat_bwd(
                _stream(dev), plan, t_plan, graph.rowptr.data_ptr(), graph.col.data_ptr(), gt.rowptr.data_ptr(), gt.col.data_ptr(),
                perm.data_ptr(), h.data_ptr(), h.stride(0), s.data_ptr(), t.data_ptr(),
                edge_scale.data_ptr() if edge_scale is not None else None, out.data_ptr(), out.stride(0), g.data_ptr(),
                g.stride(0), _dtype_code(h), rowsum.data_ptr(), rowmax.data_ptr() if rowmax is not None else None,
                dn.data_ptr(), dn.stride(0), dd.data_ptr(), grad_h.data_ptr(), grad_h.stride(0), grad_s.data_ptr(),
                grad_t.data_ptr(), graph.n_rows, graph.n_cols, heads, fo, alpha, apply_elu, mode,
                ws.data_ptr() if ws is not None else None, ws_bytes)
        _lib.check(code, "dgll_hip_gat_bwd")
        return grad_h, grad_s, grad_t, None, None, None, None, None, None, None


def gat_aggregate(graph, h, s, t, heads, alpha, apply_elu=True, mode=0, edge_scale=None, pack_scores=False):
    """Fused multi-head edge-softmax + aggregation.  h: [N, heads*fo] with fo already padded per `head_width_padded`
    (pow2=False suffices for mode 0 without edge_scale); s, t: [N, heads].  pack_scores: the caller owns the padding behind
    h's rows (h.stride(0) > h.shape[1]) and lets the kernels keep per-node scores there."""
    if not isinstance(graph, CSRGraph):
        raise TypeError("gat_aggregate expects a CSRGraph")
    fo = h.shape[1] // heads
    strided = mode == 0 and edge_scale is None
    if fo * heads != h.shape[1] or head_width_padded(fo, h.dtype, pow2=not strided) != fo:
        raise ValueError("per-head width %d is not padded for the kernels (need %d)"
                         % (fo, head_width_padded(fo, h.dtype, pow2=not strided)))
    if strided:
        return _GatAggregateStrided.apply(h, s, t, graph, heads, fo, alpha, apply_elu, pack_scores)
    return _GatAggregate.apply(h, s, t, edge_scale, graph, heads, fo, alpha, apply_elu, mode)


def gat_layer(graph, h, A, heads, alpha, apply_elu=True, pack_scores=False):
    """sparseGatConv after its transform as one autograd node: h [N, heads*fo] (per-head width a whole number of 16-byte
    vectors), A [heads*fo, 2*heads] = blockdiag(a1_k | a2_k).  GPU, mode 0, no attention dropout."""
    if not isinstance(graph, CSRGraph):
        raise TypeError("gat_layer expects a CSRGraph")
    fo = h.shape[1] // heads
    if fo * heads != h.shape[1] or head_width_padded(fo, h.dtype, pow2=False) != fo:
        raise ValueError("per-head width %d is not a whole number of 16-byte vectors" % fo)
    return _GatLayerStrided.apply(h, A, graph, heads, fo, alpha, apply_elu, pack_scores)


# ------------------------------------------------------------------------------------------------ split launches
# Building blocks of the partitioned GAT (dgll_amd/dist.py): the same kernels on the two column-halves of an adjacency.
def _ws(plan, heads, fo, dev, other_plan=None):
    n = int(_lib.lib.dgll_hip_gat_workspace_bytes(plan, heads, fo))
    if other_plan is not None:
        n = max(n, int(_lib.lib.dgll_hip_gat_workspace_bytes(other_plan, heads, fo)))
    return (torch.empty(n, dtype=torch.uint8, device=dev) if n else None), n


def gat_fwd_part(graph, h, s_rows, t_cols, out, rowsum, heads, fo, alpha, apply_elu, raw, accumulate):
    """One half of a split forward: rows of `graph` gather from `h` / `t_cols`; numerator into `out`, denominator into
    `rowsum` (raw), optionally on top of what a previous launch left there (accumulate)."""
    dev = h.device
    ws, nbytes = _ws(graph.plan(), heads, fo, dev)
    with torch.cuda.device(dev):
        code = _lib.lib.dgll_hip_gat_fwd_ex(
            _stream(dev), graph.plan(), graph.rowptr.data_ptr(), graph.col.data_ptr(), h.data_ptr(), h.stride(0),
            s_rows.data_ptr(), t_cols.data_ptr(), None, out.data_ptr(), out.stride(0), _dtype_code(h), rowsum.data_ptr(),
            graph.n_rows, heads, fo, float(alpha), int(apply_elu), ws.data_ptr() if ws is not None else None, nbytes,
            int(raw), int(accumulate))
    _lib.check(code, "dgll_hip_gat_fwd_ex")


def gat_bwd_rows_part(graph, h_cols, s_rows, t_cols, out, g, rowsum, dn, dd, grad_s, heads, fo, alpha, apply_elu, accumulate,
                      partial=None):
    """Rows pass over one column half of A.  accumulate: False / 0 = first (or only) launch (writes DN, dd, grad_s; dd_i from the
    stored output row), True / 1 = a further half (grad_s +=), 3 = DECLARED the only launch over these rows: exact dd_i;
    4 / 5 / 6 with `partial` (fp32 [n_rows, 3 * heads], the same buffer every time) = first / middle / last launch of a SPLIT
    exact pass (dgll_hip_gat_bwd_rows_split)."""
    dev = out.device
    ws, nbytes = _ws(graph.plan(), heads, fo, dev)
    with torch.cuda.device(dev):
        if int(accumulate) >= 4:
            if partial is None or partial.dtype != torch.float32 or partial.numel() < graph.n_rows * 3 * heads or not partial.is_contiguous():
                raise ValueError("a split exact rows pass needs a contiguous fp32 [n_rows, 3 * heads] `partial` buffer")
            code = _lib.lib.dgll_hip_gat_bwd_rows_split(
                _stream(dev), graph.plan(), graph.rowptr.data_ptr(), graph.col.data_ptr(), h_cols.data_ptr(), h_cols.stride(0),
                s_rows.data_ptr(), t_cols.data_ptr(), out.data_ptr(), out.stride(0), g.data_ptr(), g.stride(0), _dtype_code(out),
                rowsum.data_ptr(), dn.data_ptr(), dn.stride(0), dd.data_ptr(), grad_s.data_ptr(), graph.n_rows, heads, fo,
                float(alpha), int(apply_elu), int(accumulate), partial.data_ptr(), ws.data_ptr() if ws is not None else None, nbytes)
            _lib.check(code, "dgll_hip_gat_bwd_rows_split")
            return
        code = _lib.lib.dgll_hip_gat_bwd_rows(
            _stream(dev), graph.plan(), graph.rowptr.data_ptr(), graph.col.data_ptr(), h_cols.data_ptr(), h_cols.stride(0),
            s_rows.data_ptr(), t_cols.data_ptr(), None, out.data_ptr(), out.stride(0), g.data_ptr(), g.stride(0),
            _dtype_code(out), rowsum.data_ptr(), None, dn.data_ptr(), dn.stride(0), dd.data_ptr(), grad_s.data_ptr(),
            graph.n_rows, heads, fo, float(alpha), int(apply_elu), 0, int(accumulate), ws.data_ptr() if ws is not None else None,
            nbytes)
    _lib.check(code, "dgll_hip_gat_bwd_rows")


def gat_bwd_cols_part(graph_t, dn, h_rows, t_rows, s_cols, dd_cols, grad_h, grad_t, heads, fo, alpha):
    """Pass 2 over a transposed structure `graph_t` (rows = source nodes with features h_rows / scores t_rows, columns =
    destination rows with dn / s_cols / dd_cols)."""
    dev = dn.device
    ws, nbytes = _ws(graph_t.plan(), heads, fo, dev)
    with torch.cuda.device(dev):
        code = _lib.lib.dgll_hip_gat_bwd_cols(
            _stream(dev), graph_t.plan(), graph_t.rowptr.data_ptr(), graph_t.col.data_ptr(), None, dn.data_ptr(), dn.stride(0),
            h_rows.data_ptr(), h_rows.stride(0), t_rows.data_ptr(), s_cols.data_ptr(), dd_cols.data_ptr(), None, None,
            grad_h.data_ptr(), grad_h.stride(0), grad_t.data_ptr(), _dtype_code(dn), graph_t.n_rows, heads, fo, float(alpha), 0,
            ws.data_ptr() if ws is not None else None, nbytes)
    _lib.check(code, "dgll_hip_gat_bwd_cols")
