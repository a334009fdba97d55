"""Dense transforms that surround the aggregation (x.W of gcnconv.py:30, sageconv.py:41,72, gatconv.py:31,117).

On the GPU, for bf16 operands with 16-byte aligned rows, every product of a training step runs on the hand-written MFMA
kernels of csrc/dense.hip and csrc/gradw.hip: the forward transform (two products, bias / addend / activation fused),
both input gradients g.Ws^T, g.Wn^T from one launch (`transform_bf16_dual`), and the weight gradients x^T.g -- a reduction
over millions of rows into a tiny [F, H] output -- split over row slabs with g read once for a pair (`grad_weight_pair`).
Other dtypes / shapes / devices fall back to torch (fp32 CPU parity paths, short mini-batch blocks).
"""
import torch

from . import _lib

_SLABS = 256
_GW_MIN_ROWS = 256       # rows per slab of the split-K weight gradient below which no further slabs are made (tools/gradw_short_probe.py:
                         # 1 k / 11 k-row operands 35 / 38 -> 22 / 25 us per launch against 1024; no difference from 112 k rows up)


def _pad_wt(wt, rows=None):
    """[N, K] -> zero-padded bf16 [64 | 128 | 256 rows, 64*ceil(K/64) columns] as dgll_hip_transform_bf16 wants its
    weights (the kernel is instantiated for 2, 4 or 8 column tiles of 32 and stages that many rows).  wt: fp32 or bf16, any
    strides (a parameter's transposed view): cast, padding and layout are ONE launch (dgll_hip_pack_weight_bf16) -- the layers pass
    their fp32 parameters straight through (`wcast`).  A parameter owned by optim.FlatAdam(pack_weights=True) needs no launch at
    all: the optimizer's update kernel keeps both packed forms current and they are looked up here."""
    base = wt._base if wt._base is not None else wt
    ref = getattr(base, "_dgll_flat", None)
    if ref is not None:
        hit = ref[0].packed_for(wt)
        if hit is not None and (rows is None or hit.shape[0] == rows):
            return hit
    return _pack_now(wt, rows)


def _pack_now(wt, rows=None):
    n, k = wt.shape
    rows = rows or (64 if n <= 64 else 128 if n <= 128 else 256)
    ld = -(-k // 64) * 64
    if wt.is_cuda and wt.dtype in (torch.float32, torch.bfloat16):
        wt = wt.detach()
        out = torch.empty((rows, ld), dtype=torch.bfloat16, device=wt.device)
        with _lib.on_device(wt.device):
            code = _lib.lib.dgll_hip_pack_weight_bf16(_lib.raw_stream(wt.device), wt.data_ptr(),
                                                      _lib.F32 if wt.dtype == torch.float32 else _lib.BF16, wt.stride(0), wt.stride(1),
                                                      n, k, out.data_ptr(), ld, rows)
        _lib.check(code, "dgll_hip_pack_weight_bf16")
        return out
    out = torch.zeros((rows, ld), dtype=torch.bfloat16, device=wt.device)
    out[:n, :k] = wt
    return out


def wcast(w, like):
    """The weight in the form the products below take it: on the GPU path with bf16 activations the fp32 parameter ITSELF (the one
    launch that packs it for the MFMA kernels also casts it: no bf16 copy is made per call), otherwise w.to(like.dtype)."""
    if w.is_cuda and like.dtype == torch.bfloat16 and w.dtype == torch.float32:
        return w
    return w.to(like.dtype)


def _timed(tag, device):
    """HIP-event bracket of one launch for bench.py's tables (ops.LaunchTimer); None when no timer is active."""
    from . import ops

    timer = ops.LaunchTimer.active
    return timer.start(tag, device) if timer is not None else None


def _mfma_ok(*mats):
    return all(m is None or (m.is_cuda and m.dtype == torch.bfloat16 and m.dim() == 2 and m.stride(1) == 1
                             and m.stride(0) % 8 == 0 and m.data_ptr() % 16 == 0) for m in mats)


def bit_words(n):
    """Words per row of a sign-bit matrix over n columns: 32 columns per word, rows padded to whole 16-byte vectors."""
    return -(-(-(-n // 32)) // 4) * 4


def transform_bf16(a1, wt1, a2=None, wt2=None, relu=False, out_dtype=torch.bfloat16, n_out=None, mask=None, bias=None,
                   ld_align=None, out_gate=None, row_scale=None, addend=None, out=None, gate_bits=None, bits_out=False):
    """out[M, N] = act(a1 . wt1^T (+ a2 . wt2^T)) on the MFMA kernel.  wt*: [N, K] weights (transposed), any float
    dtype; padded here.  a*: bf16 [M, K], 16-byte aligned rows.  out_gate: bf16 [M, N]; out is zeroed where it is <= 0.
    row_scale: fp32 [M] factor on the product (before bias / activation).  addend: bf16 [M, N] added before the
    activation.  out: write into this [M, N] tensor instead of allocating.
    bits_out: also return the sign bits of the result, (out, bits) with bits int32 [M, bit_words(N)] (bit b of word w: column
    32 w + b is positive) -- written by the epilogue from the values it stores.  gate_bits: such a matrix for out_gate's values;
    the kernel reads 32 bytes of a 256-column row instead of 512 (dgll_hip_transform_bf16_bits).  Both only without mask /
    row_scale / addend and with bf16 output; gate_bits is ignored otherwise when out_gate is there too."""
    n = wt1.shape[0] if n_out is None else n_out
    m = a1.shape[0]
    bits_form = mask is None and row_scale is None and addend is None and out_dtype == torch.bfloat16
    if bits_out and not bits_form:
        raise ValueError("transform_bf16(bits_out=True): bf16 output without mask / row_scale / addend")
    if gate_bits is not None:
        if not bits_form:
            if out_gate is None:
                raise ValueError("transform_bf16(gate_bits=) without out_gate: bf16 output without mask / row_scale / addend only")
            gate_bits = None
        elif gate_bits.dtype != torch.int32 or gate_bits.shape != (m, bit_words(n)) or not gate_bits.is_contiguous():
            raise ValueError("gate_bits must be a contiguous int32 [M, bit_words(N)] matrix")
    p1 = _pad_wt(wt1)
    p2 = _pad_wt(wt2) if a2 is not None else None
    own_store = out is None       # `out`: a caller's [M, N] buffer (rows of a larger matrix, say); its padding is not ours to write
    if own_store:
        ld_align = ld_align or (8 if out_dtype == torch.bfloat16 else 4)
        ld = -(-n // ld_align) * ld_align
        store = torch.empty((m, ld), dtype=out_dtype, device=a1.device)
        out = store[:, :n] if ld != n else store
    elif out.shape != (m, n) or out.dtype != out_dtype or out.stride(1) != 1 or not out.is_cuda:
        raise ValueError("transform_bf16: `out` must be a [M, N] device tensor of the output dtype with contiguous rows")
    if bias is not None:
        bias = bias.detach().float().contiguous()
    with _lib.on_device(a1.device):
        if row_scale is not None and (row_scale.dtype != torch.float32 or row_scale.shape != (m,) or not row_scale.is_contiguous()):
            raise ValueError("row_scale must be a contiguous fp32 vector with one entry per row")
        if out_gate is not None and (out_gate.dtype != torch.bfloat16 or out_gate.shape != (m, n) or out_gate.stride(1) != 1):
            raise ValueError("out_gate must be bf16 [M, N] with contiguous rows")
        if addend is not None and (addend.dtype != torch.bfloat16 or addend.shape != (m, n) or addend.stride(1) != 1):
            raise ValueError("addend must be bf16 [M, N] with contiguous rows")
        end = _timed(("transform", m, a1.shape[1], a2.shape[1] if a2 is not None else 0, n,
                      "+".join(t for t, on in (("gate", out_gate is not None and gate_bits is None), ("gatebits", gate_bits is not None),
                                               ("signbits", bits_out), ("addend", addend is not None),
                                               ("row_scale", row_scale is not None), ("mask", mask is not None)) if on)), a1.device)
        bits = torch.empty((m, bit_words(n)), dtype=torch.int32, device=a1.device) if bits_out else None
        if bits is not None or gate_bits is not None:
            code = _lib.lib.dgll_hip_transform_bf16_bits(
                _lib.raw_stream(a1.device), a1.data_ptr(), a1.stride(0), a1.shape[1], p1.data_ptr(),
                p1.stride(0), a2.data_ptr() if a2 is not None else None, a2.stride(0) if a2 is not None else 0,
                a2.shape[1] if a2 is not None else 0, p2.data_ptr() if p2 is not None else None,
                p2.stride(0) if p2 is not None else 0, p1.shape[0], out.data_ptr(), out.stride(0), m, n,
                int(relu) | (2 if own_store else 0), bias.data_ptr() if bias is not None else None,
                out_gate.data_ptr() if out_gate is not None else None, out_gate.stride(0) if out_gate is not None else 0,
                gate_bits.data_ptr() if gate_bits is not None else None, gate_bits.stride(0) if gate_bits is not None else 0,
                bits.data_ptr() if bits is not None else None, bits.stride(0) if bits is not None else 0)
            if end is not None:
                end.record(torch.cuda.current_stream(a1.device))
            _lib.check(code, "dgll_hip_transform_bf16_bits")
            return (out, bits) if bits_out else out
        code = _lib.lib.dgll_hip_transform_bf16_add(
            _lib.raw_stream(a1.device), a1.data_ptr(), a1.stride(0), a1.shape[1], p1.data_ptr(),
            p1.stride(0), a2.data_ptr() if a2 is not None else None, a2.stride(0) if a2 is not None else 0,
            a2.shape[1] if a2 is not None else 0, p2.data_ptr() if p2 is not None else None,
            p2.stride(0) if p2 is not None else 0, p1.shape[0], mask.data_ptr() if mask is not None else None,
            mask.stride(0) if mask is not None else 0, out.data_ptr(), out.stride(0),
            _lib.BF16 if out_dtype == torch.bfloat16 else _lib.F32, m, n,
            int(relu) | (2 if own_store else 0),     # bit 1: `store` is this function's own allocation: its row padding may be written
            bias.data_ptr() if bias is not None else None,
            out_gate.data_ptr() if out_gate is not None else None, out_gate.stride(0) if out_gate is not None else 0,
            row_scale.data_ptr() if row_scale is not None else None,
            addend.data_ptr() if addend is not None else None, addend.stride(0) if addend is not None else 0)
        if end is not None:
            end.record(torch.cuda.current_stream(a1.device))
    _lib.check(code, "dgll_hip_transform_bf16")
    return out


def transform_bf16_cat(a, wt_first, wt_second):
    """[a . wt_first^T | a . wt_second^T] from ONE pass over `a`: (first, second), two [M, n] views of one bf16 buffer whose rows
    hold `first` in columns [0, n) and `second` from column hp (64 for n <= 64, else 128) on: every row of either product starts on
    a 128-byte line.  The narrowing SAGE layer's h.Wn (gathered next) and h.Ws (the aggregate lands on top of it) -- h read once
    instead of once per product (sageconv.py:72-75 computes them as two matmuls over the same `src`).  wt_*: [n <= 128, K <= 256],
    any float dtype / strides; a: bf16 [M, K], 16-byte aligned rows."""
    n, k = wt_first.shape
    if wt_second.shape != (n, k) or n > 128 or k > 256:
        raise ValueError("transform_bf16_cat: two [n <= 128, K <= 256] weight matrices of one shape")
    m = a.shape[0]
    hp = 64 if n <= 64 else 128
    rows, ld = 2 * hp, -(-k // 64) * 64
    packed = torch.empty((rows, ld), dtype=torch.bfloat16, device=a.device)
    store = torch.empty((m, rows), dtype=torch.bfloat16, device=a.device)
    with _lib.on_device(a.device):
        stream = _lib.raw_stream(a.device)
        for i, wt in enumerate((wt_first, wt_second)):           # cast, zero padding and layout of each half: one small launch
            w = wt.detach()
            code = _lib.lib.dgll_hip_pack_weight_bf16(stream, w.data_ptr(), _lib.F32 if w.dtype == torch.float32 else _lib.BF16,
                                                      w.stride(0), w.stride(1), n, k, packed.data_ptr() + i * hp * ld * 2, ld, hp)
            _lib.check(code, "dgll_hip_pack_weight_bf16")
        end = _timed(("transform", m, k, 0, hp + n, "two products"), a.device)
        code = _lib.lib.dgll_hip_transform_bf16(
            stream, a.data_ptr(), a.stride(0), k, packed.data_ptr(), ld, None, 0, 0, None, 0, rows, None, 0,
            store.data_ptr(), rows, _lib.BF16, m, hp + n, 2, None)          # relu bit 1: the row padding is ours (zeros)
        if end is not None:
            end.record(torch.cuda.current_stream(a.device))
    _lib.check(code, "dgll_hip_transform_bf16")
    return store[:, :n], store[:, hp:hp + n]


def fused_ok(graph, x, n_out, h_self=None):
    """Shapes / layouts the fused aggregate -> transform kernel (csrc/fused_sage.hip) takes."""
    return (_mfma_ok(x, h_self) and x.shape[1] <= 256 and n_out <= 256 and getattr(graph, "is_cuda", False)
            and (graph.val is None or graph.val.dtype == torch.float32) and x.shape[0] == graph.n_cols
            and (h_self is None or h_self.shape[0] == graph.n_rows))


def sage_fused_forward(graph, x, reduce, h_self, wt_self, wt_nbr, relu, bias=None, keep_agg=False, ld_align=None):
    """(out, agg) with out = act(h_self . wt_self^T + reduce_A(x) . wt_nbr^T + bias) from ONE launch: a workgroup aggregates a
    32-row tile into LDS and feeds it to the MFMAs (dgll_hip_sage_fused_forward).  wt_*: [N, K] (transposed weights, any float
    dtype); wt_nbr None: the aggregate is added instead of transformed (x.shape[1] == N).  agg: the aggregated rows
    [n_rows, feat] bf16 when keep_agg (the weight gradient needs them), else None."""
    from . import ops

    dev = x.device
    feat = x.shape[1]
    n = wt_nbr.shape[0] if wt_nbr is not None else (wt_self.shape[0] if wt_self is not None else feat)
    p1 = _pad_wt(wt_self) if h_self is not None else None
    p2 = _pad_wt(wt_nbr) if wt_nbr is not None else None
    w_rows = (p1 if p1 is not None else p2).shape[0]
    if p1 is not None and p2 is not None and p1.shape[0] != p2.shape[0]:
        raise ValueError("both weight matrices must have N rows")
    ld_align = ld_align or 8
    ld = -(-n // ld_align) * ld_align
    store = torch.empty((graph.n_rows, ld), dtype=torch.bfloat16, device=dev)
    out = store[:, :n] if ld != n else store
    plan = graph.plan()
    long_rows = graph.num_long_rows() > 0
    agg = None
    if keep_agg or long_rows:
        agg = ops.alloc_features(graph.n_rows, feat, torch.bfloat16, dev)
    ws_bytes = graph.workspace_bytes(feat) if long_rows else 0
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
    if bias is not None:
        bias = bias.detach().float().contiguous()
    val = graph.val
    with torch.cuda.device(dev):
        end = _timed(("fused_sage", graph.n_rows, feat, h_self.shape[1] if h_self is not None else 0, n, graph.nnz), dev)
        code = _lib.lib.dgll_hip_sage_fused_forward(
            torch.cuda.current_stream(dev).cuda_stream, plan, graph.rowptr.data_ptr(), graph.col.data_ptr(),
            val.data_ptr() if val is not None else None, x.data_ptr(), x.stride(0), feat,
            _lib.REDUCE_MEAN if reduce == "mean" else _lib.REDUCE_SUM,
            h_self.data_ptr() if h_self is not None else None, h_self.stride(0) if h_self is not None else 0,
            h_self.shape[1] if h_self is not None else 0, p1.data_ptr() if p1 is not None else None,
            p1.stride(0) if p1 is not None else 0, p2.data_ptr() if p2 is not None else None, p2.stride(0) if p2 is not None else 0,
            w_rows, bias.data_ptr() if bias is not None else None, int(relu), out.data_ptr(), out.stride(0), n,
            agg.data_ptr() if agg is not None else None, agg.stride(0) if agg is not None else 0, graph.n_rows, graph.n_cols,
            ws.data_ptr() if ws is not None else None, ws_bytes)
        if end is not None:
            end.record(torch.cuda.current_stream(dev))
    _lib.check(code, "dgll_hip_sage_fused_forward")
    return out, (agg if keep_agg else None)


def transform_bf16_dual(a, wt1, wt2, out1=None):
    """(a . wt1^T, a . wt2^T) in one MFMA launch that reads `a` once: the two input gradients g.Ws^T, g.Wn^T of a SAGE
    layer.  wt1, wt2: [N, K] with N, K <= 256; a: bf16 [M, K], 16-byte aligned rows."""
    n, m = wt1.shape[0], a.shape[0]
    if wt2.shape != wt1.shape or n > 256 or a.shape[1] > 256:
        raise ValueError("transform_bf16_dual: two [N <= 256, K <= 256] weight matrices of one shape")
    p1, p2 = _pad_wt(wt1, rows=256), _pad_wt(wt2, rows=256)
    ld = -(-n // 8) * 8
    outs = [torch.empty((m, ld), dtype=torch.bfloat16, device=a.device) for _ in range(2)]
    o1, o2 = (o[:, :n] if ld != n else o for o in outs)
    if out1 is not None:       # the first product into a caller's rows (a range of a row_slices gradient buffer): bf16, 16-byte pitch
        if out1.shape != (m, n) or out1.dtype != torch.bfloat16 or out1.stride(1) != 1 or out1.stride(0) % 8 or out1.data_ptr() % 16:
            raise ValueError("transform_bf16_dual(out1=): [M, N] bf16 rows on a 16-byte pitch")
        o1 = out1
    with _lib.on_device(a.device):
        end = _timed(("transform_dual", m, a.shape[1], 0, 2 * n, ""), a.device)
        code = _lib.lib.dgll_hip_transform_bf16_dual(
            _lib.raw_stream(a.device), a.data_ptr(), a.stride(0), a.shape[1], p1.data_ptr(), p2.data_ptr(),
            p1.stride(0), p1.shape[0], o1.data_ptr(), o1.stride(0), o2.data_ptr(), o2.stride(0), m, n)
        if end is not None:
            end.record(torch.cuda.current_stream(a.device))
    _lib.check(code, "dgll_hip_transform_bf16_dual")
    return o1, o2


def _rows16_ok(x):
    epv = 16 // x.element_size()
    return x.dim() == 2 and x.stride(1) == 1 and x.stride(0) >= x.shape[1] and x.stride(0) % epv == 0 and x.data_ptr() % 16 == 0


def _relu_backward_rows(g, out):
    """g where out > 0 (aten::threshold_backward), ONE pass whatever g's row pitch.  A gradient whose rows already start on 16-byte
    boundaries (the loss's: 41 classes on a 128-byte pitch) keeps that layout -- the masked copy is written into a buffer of the same
    pitch, so the products that follow read it as it is; made dense first (`.contiguous()`) its 82-byte rows were re-laid out four times
    (twice for the input gradients, twice for the weight gradients: five tiny launches per mini-batch step)."""
    if g.dim() == 2 and g.stride(1) == 1 and not g.is_contiguous() and _rows16_ok(g):
        buf = torch.empty((g.shape[0], g.stride(0)), dtype=g.dtype, device=g.device)
        gm = buf[:, :g.shape[1]]
        torch.ops.aten.threshold_backward.grad_input(g, out, 0, grad_input=gm)
        return gm
    return torch.ops.aten.threshold_backward(g.contiguous(), out, 0)


def _as_rows16(x):
    """x with unit column stride and 16-byte aligned rows: itself when it already is, else one copy into a padded buffer."""
    epv = 16 // x.element_size()
    if (x.dim() == 2 and x.stride(1) == 1 and x.stride(0) >= x.shape[1] and x.stride(0) % epv == 0 and x.data_ptr() % 16 == 0):
        return x
    ld = -(-x.shape[1] // epv) * epv
    buf = torch.empty((x.shape[0], ld), dtype=x.dtype, device=x.device)
    out = buf[:, :x.shape[1]] if ld != x.shape[1] else buf
    out.copy_(x)
    return out


def _mm_f32(a, wt, relu, bias, addend, a2=None, wt2=None, gate=None):
    """fp32 product(s) on dgll_hip_mm2_f32 (v_mfma_f32_32x32x2_f32: exact fp32 arithmetic), 256 output columns per launch:
    gate(act(a . wt^T (+ a2 . wt2^T) + addend + bias)), both products in one accumulation."""
    a = a if a.stride(1) == 1 else a.contiguous()
    wt = wt.to(torch.float32)
    wt = wt if wt.stride(1) == 1 else wt.contiguous()
    m, k = a.shape
    n = wt.shape[0]
    k2 = 0
    if a2 is not None:
        a2 = a2.to(torch.float32)
        a2 = a2 if a2.stride(1) == 1 else a2.contiguous()
        wt2 = wt2.to(torch.float32)
        wt2 = wt2 if wt2.stride(1) == 1 else wt2.contiguous()
        k2 = a2.shape[1]
    out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    if addend is not None:
        addend = addend.to(torch.float32)
        addend = addend if addend.stride(1) == 1 else addend.contiguous()
    if gate is not None:
        gate = gate.to(torch.float32)
        gate = gate if gate.stride(1) == 1 else gate.contiguous()
    if bias is not None:
        bias = bias.detach().to(torch.float32).contiguous()
    with _lib.on_device(a.device):
        stream = _lib.raw_stream(a.device)
        end = _timed(("transform_f32", m, k, k2, n, "+".join(t for t, on in (("gate", gate is not None), ("addend", addend is not None)) if on)),
                     a.device)
        for n0 in range(0, n, 256):
            nn = min(256, n - n0)
            code = _lib.lib.dgll_hip_mm2_f32(
                stream, a.data_ptr(), a.stride(0), wt.data_ptr() + n0 * wt.stride(0) * 4, wt.stride(0), k,
                a2.data_ptr() if a2 is not None else None, a2.stride(0) if a2 is not None else 0,
                wt2.data_ptr() + n0 * wt2.stride(0) * 4 if a2 is not None else None, wt2.stride(0) if a2 is not None else 0, k2,
                out.data_ptr() + n0 * 4, out.stride(0), m, nn,
                bias.data_ptr() + n0 * 4 if bias is not None else None, int(relu),
                addend.data_ptr() + n0 * 4 if addend is not None else None, addend.stride(0) if addend is not None else 0,
                gate.data_ptr() + n0 * 4 if gate is not None else None, gate.stride(0) if gate is not None else 0)
            _lib.check(code, "dgll_hip_mm2_f32")
        if end is not None:
            end.record(torch.cuda.current_stream(a.device))
    return out


def mm_nt(a, wt, relu=False, bias=None, addend=None, out=None):
    """act(a . wt^T + addend + bias) for a [M, K], wt [N, K] on the hand-written kernels: bf16 on the MFMA transform (rows are
    re-laid out to 16-byte alignment when they are not; more than 256 output columns run as 256-column launches), fp32 on the
    fp32 matrix-core kernel.  Host tensors (the layers' CPU logic) use torch."""
    if not a.is_cuda:
        out = a @ wt.t().to(a.dtype)
        if addend is not None:
            out = out + addend
        if bias is not None:
            out = out + bias
        return torch.relu(out) if relu else out
    if a.dtype == torch.float32:
        return _mm_f32(a, wt, relu, bias, addend)
    if a.dtype != torch.bfloat16:
        raise TypeError("dgll_amd dense products take float32 or bfloat16 matrices, got %s" % a.dtype)
    a = _as_rows16(a)
    if addend is not None:
        addend = addend.to(torch.bfloat16)
        addend = addend if addend.stride(1) == 1 else addend.contiguous()
    n = wt.shape[0]
    if n <= 256:
        return transform_bf16(a, wt, relu=relu, bias=bias, addend=addend, out=out)
    parts = [transform_bf16(a, wt[n0:n0 + 256], relu=relu, bias=None if bias is None else bias[n0:n0 + 256],
                            addend=None if addend is None else addend[:, n0:n0 + 256]) for n0 in range(0, n, 256)]
    return torch.cat(parts, dim=1)


def mm2_nt(a1, wt1, a2, wt2, relu=False, gate=None):
    """gate(act(a1 . wt1^T + a2 . wt2^T)): both products accumulate in fp32 before anything is stored; gate: [M, N], the result is
    zeroed where it is <= 0 (the ReLU mask of the layer below, fused for fp32 GPU tensors, a threshold pass otherwise).  bf16 operands whose rows are
    not 16-byte aligned are re-laid out once and take the two-product MFMA launch (256 output columns at a time) -- the sum
    of two separately stored bf16 products rounds the first one to 8 bits before the add, which showed as a 4-5 % relative
    error of the first layer's gradients at the Reddit shape (602-column rows; tests/test_config2_reddit_gpu.py)."""
    if a1.is_cuda and a1.dtype == torch.float32 and a2.dtype == torch.float32:
        return _mm_f32(a1, wt1, relu, None, None, a2=a2, wt2=wt2, gate=gate)       # one launch, one accumulation (exact fp32 FMAs)
    if not (a1.is_cuda and a1.dtype == torch.bfloat16 and a2.dtype == torch.bfloat16):
        res = mm_nt(a2, wt2, relu=relu, addend=mm_nt(a1, wt1))      # mixed / host: the addend stays fp32 (exact); host: torch
        return res if gate is None else torch.ops.aten.threshold_backward(res, gate.to(res.dtype), 0)
    a1, a2 = _as_rows16(a1), _as_rows16(a2)
    n = wt1.shape[0]
    parts = [transform_bf16(a1, wt1[n0:n0 + 256], a2, wt2[n0:n0 + 256], relu=relu) for n0 in range(0, n, 256)]
    res = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
    return res if gate is None else torch.ops.aten.threshold_backward(res, gate.to(res.dtype), 0)


def input_grads(g, wsd, wnd, out1=None):
    """(g . Ws^T, g . Wn^T) for weights stored [in, out]: one MFMA launch that reads g once for tall bf16 gradients, else two.
    out1: optional destination rows of the first product (transform_bf16_dual / mm_nt's `out`)."""
    if g.shape[0] >= 64 * _SLABS and _mfma_ok(g) and wsd.shape == wnd.shape and max(wsd.shape) <= 256:
        if out1 is not None and (out1.dtype != torch.bfloat16 or out1.stride(0) % 8 or out1.data_ptr() % 16):
            out1 = None
        return transform_bf16_dual(g, wsd, wnd, out1=out1)
    # (short gradients: two launches; the first product still lands in the caller's rows when the kernel can write there)
    return input_grad(g, wsd, out=out1) if out1 is not None else mm_nt(g, wsd), mm_nt(g, wnd)


def _rows(g):
    """g with contiguous rows, keeping a padded leading dimension (a `.contiguous()` would squeeze a 16-byte aligned
    [N, 47] view of a [N, 48] buffer into unaligned 94-byte rows and push the products below off the MFMA kernels)."""
    return g if g.dim() == 2 and g.stride(1) == 1 else g.contiguous()


def input_grad(g, wd, out=None):
    """g . W^T for a weight stored [in, out] (the transposed weight of this product is W itself).  out: bf16 GPU products of at most
    256 columns are written straight into it; anything else is computed and copied."""
    if out is not None and g.is_cuda and g.dtype == torch.bfloat16 and wd.shape[0] <= 256 and out.dtype == torch.bfloat16:
        return mm_nt(g, wd, out=out)
    res = mm_nt(g, wd)
    if out is not None:
        out.copy_(res)
        return out
    return res


_GW_WORKSPACE = {}


def _gradw_ok(*mats):
    return all(m is not None and m.is_cuda and m.dtype == torch.bfloat16 and m.dim() == 2 and m.stride(1) == 1
               and m.stride(0) % 8 == 0 and m.data_ptr() % 16 == 0 and m.shape[1] <= 256 for m in mats)


def _out_ok(out, k, n, device):
    return (out is not None and out.dtype == torch.float32 and out.shape == (k, n) and out.stride(1) == 1 and out.stride(0) >= n
            and out.device == device)


def _grad_weight_hip(x1, x2, g, out1=None, out2=None, transposed=False):
    """(x1^T . g, x2^T . g) fp32 on the split-K MFMA kernel (csrc/gradw.hip): g is read once for both products.  out1 / out2:
    fp32 [K, N] destinations (a parameter's slot of optim.FlatAdam's gradient buffer) written instead of fresh tensors.
    transposed: the results are stored as (g^T . x1, g^T . x2), [N, K] each (dgll_hip_grad_weight_bf16_tr)."""
    m, n = g.shape
    k1, k2 = x1.shape[1], (x2.shape[1] if x2 is not None else 0)
    types = (-(-k1 // 64) + -(-k2 // 64) + 3) // 4
    n_cu = torch.cuda.get_device_properties(g.device).multi_processor_count
    slabs = max(1, min(4096, n_cu // types, -(-m // _GW_MIN_ROWS)))   # at least _GW_MIN_ROWS rows per slab: short operands (sampled
                                                                       # blocks) would otherwise pay for summing hundreds of near-empty partials
    ld_max = max(x1.stride(0), x2.stride(0) if x2 is not None else 0, g.stride(0))

    def slab_fits(sl):      # the kernel's own check (gradw.hip): rows per slab rounded up to 64, + 256 rows of prefetch overshoot, in uint32 bytes
        per = max(64, -(-(-(-m // sl)) // 64) * 64)
        return (per + 256) * ld_max * 2 < (1 << 32)

    while slabs < 4096 and not slab_fits(slabs):
        slabs *= 2
    slabs = min(slabs, 4096)
    if not slab_fits(slabs):
        raise ValueError("grad_weight: rows of %d elements are too wide for the split-K kernel's 32-bit slab offsets even at 4096 slabs "
                         "(%d rows): split the reduction over row blocks" % (ld_max, m))
    need = int(_lib.lib.dgll_hip_grad_weight_workspace(k1, k2, slabs))
    key = (g.device.index, _lib.raw_stream(g.device))
    ws = _GW_WORKSPACE.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = _GW_WORKSPACE[key] = torch.empty(need // 4, dtype=torch.float32, device=g.device)
    s1, s2 = ((n, k1), (n, k2)) if transposed else ((k1, n), (k2, n))
    d1 = out1 if _out_ok(out1, s1[0], s1[1], g.device) else torch.empty(s1, dtype=torch.float32, device=g.device)
    d2 = (out2 if _out_ok(out2, s2[0], s2[1], g.device) else torch.empty(s2, dtype=torch.float32, device=g.device)) if x2 is not None else None
    entry = "dgll_hip_grad_weight_bf16_tr" if transposed else "dgll_hip_grad_weight_bf16"
    with _lib.on_device(g.device):       # the launch (and its event bracket) belong to g's device, whatever is current
        end = _timed(("grad_weight", m, k1, k2, n, "tr" if transposed else ""), g.device)
        code = getattr(_lib.lib, entry)(
            _lib.raw_stream(g.device), x1.data_ptr(), x1.stride(0), k1,
            x2.data_ptr() if x2 is not None else None, x2.stride(0) if x2 is not None else 0, k2, g.data_ptr(), g.stride(0), n, m,
            ws.data_ptr(), ws.numel() * 4, slabs, d1.data_ptr(), d1.stride(0), d2.data_ptr() if d2 is not None else None,
            d2.stride(0) if d2 is not None else 0)
        if end is not None:
            end.record(torch.cuda.current_stream(g.device))
    _lib.check(code, entry)
    return d1, d2


def _grad_weight_f32(x, g):
    """fp32 x^T . g on dgll_hip_grad_weight_f32 (slab partials summed in slab order)."""
    x = x.to(torch.float32)
    g = g.to(torch.float32)
    x = x if x.stride(1) == 1 else x.contiguous()
    g = g if g.stride(1) == 1 else g.contiguous()
    m, k = x.shape
    n = g.shape[1]
    if m == 0:                                   # an empty reduction (an empty tensor has no storage to point the kernel at)
        return torch.zeros((k, n), dtype=torch.float32, device=x.device)
    # row slabs: enough workgroups (k-blocks x n-blocks x slabs) for two per CU, at least 512 rows each
    blocks = -(-k // 256) * -(-n // 256)
    n_cu = torch.cuda.get_device_properties(x.device).multi_processor_count
    slabs = max(1, min(-(-2 * n_cu // blocks), -(-m // 512)))
    need = int(_lib.lib.dgll_hip_grad_weight_f32_workspace(k, n, slabs))
    ws = torch.empty(need // 4, dtype=torch.float32, device=x.device)
    out = torch.empty((k, n), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        end = _timed(("grad_weight_f32", m, k, 0, n, ""), x.device)
        code = _lib.lib.dgll_hip_grad_weight_f32(torch.cuda.current_stream(x.device).cuda_stream, x.data_ptr(), x.stride(0),
                                                 g.data_ptr(), g.stride(0), out.data_ptr(), out.stride(0), m, k, n, ws.data_ptr(),
                                                 need, slabs)
        if end is not None:
            end.record(torch.cuda.current_stream(x.device))
    _lib.check(code, "dgll_hip_grad_weight_f32")
    return out


def _into(out, res):
    if out is None or out is res:
        return res
    if out.shape != res.shape:
        return res
    out.copy_(res)
    return out


def grad_weight(x, g, out=None):
    """x^T . g  for x [M, K], g [M, N] -> fp32 [K, N]: the split-K MFMA kernel for bf16 (operands re-laid out to 16-byte rows
    when needed, 256 x 256 output blocks), the fp32 slab kernel for fp32; host tensors use torch.  out: fp32 [K, N] destination
    (optim.grad_slot_of(parameter)): written by the kernel itself where it can be, copied into otherwise."""
    if not x.is_cuda:
        return _into(out, torch.mm(x.t(), g.to(x.dtype)).float())
    if x.dtype == torch.float32 or g.dtype == torch.float32:
        return _into(out, _grad_weight_f32(x, g))
    x, g = _as_rows16(x), _as_rows16(g.to(x.dtype))
    k, n = x.shape[1], g.shape[1]
    if k <= 256 and n <= 256:
        return _grad_weight_hip(x, None, g, out1=out)[0]
    if out is not None and _out_ok(out, k, n, x.device):
        res = grad_weight(x, g)
        out.copy_(res)
        return out
    out = torch.empty((k, n), dtype=torch.float32, device=x.device)
    for n0 in range(0, n, 256):
        gs = g[:, n0:n0 + 256]
        for k0 in range(0, k, 512):          # two 256-column blocks of x per launch: g is read once for both
            x1 = x[:, k0:k0 + 256]
            x2 = x[:, k0 + 256:k0 + 512] if k0 + 256 < k else None
            d1, d2 = _grad_weight_hip(x1, x2, gs)
            out[k0:k0 + 256, n0:n0 + 256] = d1
            if d2 is not None:
                out[k0 + 256:k0 + 512, n0:n0 + 256] = d2
    return out


def grad_weight_pair(x1, x2, g, out1=None, out2=None):
    """(x1^T . g, x2^T . g): the two weight gradients of a SAGE layer share g -- one launch reads it once.  Operands wider than 256
    columns (the 602-column first layer of the Reddit shape) are cut into 256-column blocks and the blocks of BOTH operands are
    paired two per launch: 3 launches for 2 x 602 columns where one operand after the other took 4."""
    if _gradw_ok(x1, x2, g):
        return _grad_weight_hip(x1, x2, g, out1=out1, out2=out2)
    wide = (x1.is_cuda and x1.dtype == x2.dtype == g.dtype == torch.bfloat16 and g.shape[1] <= 256 and _gradw_ok(g)
            and max(x1.shape[1], x2.shape[1]) > 256)
    if wide:
        x1, x2 = _as_rows16(x1), _as_rows16(x2)
        n = g.shape[1]
        outs = [o if _out_ok(o, x.shape[1], n, g.device) and o.is_contiguous() else torch.empty((x.shape[1], n), dtype=torch.float32, device=g.device)
                for o, x in ((out1, x1), (out2, x2))]
        blocks = [(x[:, k0:k0 + 256], o[k0:k0 + 256]) for x, o in zip((x1, x2), outs) for k0 in range(0, x.shape[1], 256)]
        if all(_gradw_ok(b) for b, _ in blocks):
            for i in range(0, len(blocks), 2):
                (xa, oa), (xb, ob) = blocks[i], (blocks[i + 1] if i + 1 < len(blocks) else (None, None))
                _grad_weight_hip(xa, xb, g, out1=oa, out2=ob)
            return outs[0], outs[1]
    return grad_weight(x1, g, out=out1), grad_weight(x2, g, out=out2)


def grad_weight_shared_x(x, g1, g2, out1=None, out2=None):
    """(x^T . g1, x^T . g2): the two weight gradients of the narrowing SAGE layer share the WIDE operand x (fused_layers.
    _SageGraphLayerTransformFirst: g1 the output gradient, g2 its transposed aggregation).  One launch with the roles swapped and
    the results stored transposed reads x once: two launches each read it whole (2 x 1.25 GB of the headline step)."""
    if _gradw_ok(g1, g2, x) and g1.shape[1] <= 256 and g2.shape[1] <= 256 and x.shape[1] <= 256:
        return _grad_weight_hip(g1, g2, x, out1=out1, out2=out2, transposed=True)
    return grad_weight(x, g1, out=out1), grad_weight(x, g2, out=out2)


def column_sum(g, slabs=2048):
    """g.sum(0) in fp32 for a tall matrix (bias gradients).  torch's single-pass column reduction of a [2.4 M, 47] bf16
    matrix takes 12-18 ms; reduced in two stages -- [slabs, rows, F] over rows, then over slabs -- it reads g once at HBM
    speed (0.10 ms), with fp32 accumulation throughout.  (A ones-vector GEMM costs 1-2 ms; the batched split-K form spends
    11 ms per call on the host in the library's heuristics for that shape.)"""
    m = g.shape[0]
    if m < 64 * slabs:
        return g.sum(0, dtype=torch.float32)
    rows = m // slabs
    main = rows * slabs
    out = g[:main].view(slabs, rows, g.shape[1]).sum(1, dtype=torch.float32).sum(0)
    if main < m:
        out = out + g[main:].sum(0, dtype=torch.float32)
    return out


class _SageTransform(torch.autograd.Function):
    """relu?( h.Ws + agg.Wn ) with one fused add (addmm) and in-place activation; backward shares the masked gradient
    between the four products (sageconv.py:71-82 computes the same terms as separate autograd nodes)."""

    @staticmethod
    def forward(ctx, h, agg, ws, wn, relu):
        wsd, wnd = wcast(ws, h), wcast(wn, h)
        ctx.mfma = _mfma_ok(h, agg) and ws.shape[1] <= 256
        if ctx.mfma:   # one MFMA launch: both products, the add and the ReLU, every activation row read once
            out = transform_bf16(h, wsd.t(), agg, wnd.t(), relu=relu)
        else:
            out = mm2_nt(h, wsd.t(), agg, wnd.t(), relu=relu)
        ctx.relu = relu
        ctx.wparams = (ws, wn)
        ctx.h_dest = getattr(h, "_dgll_grad_dest", None)      # h is a row range of a stacked layer input (ops.row_slices)
        ctx.save_for_backward(h, agg, wsd, wnd, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, g):
        from .ops import grad_dest
        from .optim import grad_slot_of

        h, agg, wsd, wnd, out = ctx.saved_tensors
        if ctx.relu:
            g = _relu_backward_rows(g, out)   # one vectorised pass: g where out > 0
        elif not (g.stride(1) == 1 and _rows16_ok(g)):
            g = g.contiguous()
        if ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            # the self path's product straight into its rows of the stacked input's gradient buffer (no copy pass afterwards)
            got = grad_dest(ctx.h_dest, h.shape, g.dtype, g.device, allow_accumulate=False) if (ctx.mfma and g.is_cuda) else None
            gh, gagg = input_grads(g, wsd, wnd, out1=got[0] if got is not None else None)
            if got is not None and gh.data_ptr() != got[0].data_ptr():
                got[0].copy_(gh)                      # (the product took another path: honour the claim)
                gh = got[0]
        else:
            gh = mm_nt(g, wsd) if ctx.needs_input_grad[0] else None
            gagg = mm_nt(g, wnd) if ctx.needs_input_grad[1] else None
        if ctx.needs_input_grad[2] and ctx.needs_input_grad[3]:
            gws, gwn = grad_weight_pair(h, agg, g, out1=grad_slot_of(ctx.wparams[0]), out2=grad_slot_of(ctx.wparams[1]))
        else:
            gws = grad_weight(h, g, out=grad_slot_of(ctx.wparams[0])) if ctx.needs_input_grad[2] else None
            gwn = grad_weight(agg, g, out=grad_slot_of(ctx.wparams[1])) if ctx.needs_input_grad[3] else None
        return gh, gagg, gws, gwn, None


def sage_transform(h, agg, ws, wn, relu):
    return _SageTransform.apply(h, agg, ws, wn, relu)


class _Linear(torch.autograd.Function):
    """x . w, g . w^T and x^T . g on the hand-written kernels (bf16 MFMA transform / split-K, fp32 matrix-core kernels)."""

    @staticmethod
    def forward(ctx, x, w):
        wd = wcast(w, x)
        if _mfma_ok(x) and wd.shape[1] <= 256:
            # narrow outputs feed a gather: give every row its own 128-byte line (a 94-byte row would straddle two)
            out = transform_bf16(x, wd.t(), ld_align=64 if wd.shape[1] < 64 else None)
        else:
            out = mm_nt(x, wd.t())
        ctx.save_for_backward(x, wd)
        return out

    @staticmethod
    def backward(ctx, g):
        x, wd = ctx.saved_tensors
        g = _rows(g)
        gx = input_grad(g, wd) if ctx.needs_input_grad[0] else None
        gw = grad_weight(x, g) if ctx.needs_input_grad[1] else None
        return gx, gw


def linear(x, w):
    return _Linear.apply(x, w)


class _SkinnyLinear(torch.autograd.Function):
    """x[M, K] . w[K, n] -> fp32 [M, n] for a handful of output columns (the per-node attention scores of all GAT heads:
    n = 2 * heads).  Forward on the MFMA kernel with fp32 output (no bf16 rounding of the scores); the weight gradient
    x^T . g -- an M = millions reduction into a K x n sliver, for which the library picks a 5 ms kernel at the products
    shape -- goes through the split-K batched GEMM (0.3 ms)."""

    @staticmethod
    def forward(ctx, x, w):
        wd = wcast(w, x)
        if _mfma_ok(x) and wd.shape[1] <= 256:
            out = transform_bf16(x, wd.t(), out_dtype=torch.float32)
        else:
            out = mm_nt(x, wd.t()).float()
        ctx.save_for_backward(x, wd)
        return out

    @staticmethod
    def backward(ctx, g):
        x, wd = ctx.saved_tensors
        n = g.shape[1]
        if n % 8 and _mfma_ok(x) and x.shape[0] >= 64 * _SLABS:
            # a handful of columns: rows of 2 * heads bf16 are not 16-byte aligned.  Zero-padded to 8 columns both products
            # run on the MFMA kernels (the extra [M, 8] matrix is noise next to x) instead of library GEMMs
            pad = -n % 8
            gd = torch.zeros((g.shape[0], n + pad), dtype=x.dtype, device=g.device)
            gd[:, :n] = g
            wd = torch.nn.functional.pad(wd, (0, pad))
        else:
            gd = g.to(x.dtype).contiguous()
        gx = input_grad(gd, wd) if ctx.needs_input_grad[0] else None
        gw = grad_weight(x, gd)[:, :n] if ctx.needs_input_grad[1] else None
        return gx, gw


def skinny_linear(x, w):
    return _SkinnyLinear.apply(x, w)


class _AddLinearAct(torch.autograd.Function):
    """act(addend + x . w): the self term of a transform-first SAGE layer (the neighbour term arrives already
    aggregated).  One MFMA launch forward (addend in the epilogue); MFMA input gradient and split-K weight gradient."""

    @staticmethod
    def forward(ctx, addend, x, w, relu):
        wd = wcast(w, x)
        if _mfma_ok(x) and addend.dtype == torch.bfloat16 and addend.stride(1) == 1 and wd.shape[1] <= 256:
            out = transform_bf16(x, wd.t(), relu=relu, addend=addend)      # one MFMA launch instead of addmm + ReLU pass
        else:
            out = mm_nt(x, wd.t(), relu=relu, addend=addend)
        ctx.relu = relu
        ctx.save_for_backward(x, wd, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, g):
        x, wd, out = ctx.saved_tensors
        if ctx.relu:
            g = torch.ops.aten.threshold_backward(g.contiguous(), out, 0)
        g = g.contiguous()
        gx = input_grad(g, wd) if ctx.needs_input_grad[1] else None
        gw = grad_weight(x, g) if ctx.needs_input_grad[2] else None
        return (g if ctx.needs_input_grad[0] else None), gx, gw, None


def add_linear_act(addend, x, w, relu):
    return _AddLinearAct.apply(addend, x, w, relu)
