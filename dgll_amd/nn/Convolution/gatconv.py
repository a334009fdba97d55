"""GAT layers on the gfx950 aggregation engine.

Interface mirror of /root/reference/dgll/nn/Convolution/gatconv.py:
  gatConv(in_features, out_features, dropout, alpha, concat=True)          dense-adjacency GAT   gatconv.py:10-57
  SpecialSpmmFunction / SpecialSpmm                                        sparse A.b with grads gatconv.py:60-86
  sparseGatConv(in_features, out_features, dropout, alpha, concat=True)    sparse GAT            gatconv.py:89-151
  GAT / SpGAT(nfeat, nhid, nclass, dropout, alpha, nheads)                 8-head models         gatconv.py:154-199
with identical parameter names (`W`, `a`, `attention_%d`, `out_att`) and initialisers.

On the GPU each layer is ONE fused kernel (dgll_hip_gat_fwd): edge scores are formed from two per-node dot
products s_i = a[:fo].h_i, t_j = a[fo:].h_j instead of the reference's materialised [2*fo, E] edge_h
(gatconv.py:122), all heads of SpGAT/GAT run in the same launch (the reference calls heads one by one,
gatconv.py:168,196), and the backward pass is two more gather passes instead of a dense N x N matmul
(gatconv.py:76).  The adjacency may be the reference's dense 0/1 matrix (its nonzero pattern is converted to
CSR once and cached), a torch sparse tensor or a CSRGraph.

Not reproduced: the host-synchronising `assert not isnan(...)` checks (gatconv.py:119,126,136,141; set
DGLL_CHECK_NAN=1 to get them back), and gatConv's behaviour for rows with no edge (the reference's dense
softmax then attends uniformly to ALL nodes; here, as in sparseGatConv, such rows are NaN -- add self-loops).
"""
import os

from ... import backend as F
from ... import dense, ops
from ...graph import CSRGraph, as_csr_graph

_CHECK_NAN = os.environ.get("DGLL_CHECK_NAN", "0") == "1"


class _UnpadOneHead(F.autograd.Function):
    """out[:, :fo] of a one-head layer whose rows were computed fo_pad wide (47 classes in 48 columns).  As plain tensor ops the
    backward is a zero-filled [N, fo_pad] buffer and a strided 2-byte copy of the gradient into it (0.2 ms at the products size).  When
    the gradient that arrives is itself a [N, fo] view of rows whose padding up to fo_pad is known to be ZERO -- ops.cross_entropy
    writes its gradient rows that way and says so (`_dgll_zero_padding`) -- the padded gradient is that very buffer, re-viewed."""

    @staticmethod
    def forward(ctx, out, fo, fo_pad):
        ctx.cfg = (fo, fo_pad)
        return out[:, :fo]

    @staticmethod
    def backward(ctx, g):
        fo, fo_pad = ctx.cfg
        tag = getattr(g, "_dgll_zero_padding", None)
        if (tag is not None and tag[1] == g.data_ptr() and tag[2] == g._version and tag[0] >= fo_pad and g.dim() == 2 and g.shape[1] == fo
                and g.stride(1) == 1 and g.stride(0) >= fo_pad):
            return g.as_strided((g.shape[0], fo_pad), (g.stride(0), 1), g.storage_offset()), None, None
        buf = g.new_zeros((g.shape[0], fo_pad))
        buf[:, :fo] = g
        return buf, None, None


def _unpad_heads(out, heads, fo, fo_pad):
    if fo_pad == fo:
        return out
    if heads == 1 and out.is_cuda:
        return _UnpadOneHead.apply(out, fo, fo_pad)
    return out.view(out.shape[0], heads, fo_pad)[:, :, :fo].reshape(out.shape[0], heads * fo)


_PACK_INDEX = {}


def _pack_index(heads, fo, fo_pad, device):
    """(row, column) of every element of the stacked [heads, 2 fo] attention vectors inside the block-diagonal score matrix
    A [heads * fo_pad, 2 heads]: a1 of head k fills column k, a2 column heads + k, rows k fo_pad .. k fo_pad + fo."""
    key = (heads, fo, fo_pad, str(device))
    idx = _PACK_INDEX.get(key)
    if idx is None:
        k = F.arange(heads, device=device).repeat_interleave(2 * fo)
        j = F.arange(2 * fo, device=device).repeat(heads)
        second = (j >= fo).long()
        idx = _PACK_INDEX[key] = (k * fo_pad + j - second * fo, k + second * heads)
    return idx


class _PackHeads(F.autograd.Function):
    """The per-head parameters of a multi-head layer as the two matrices its launches take: W [Fin, heads * fo_pad] (every head's
    transform side by side, gatconv.py:196's loop as one product) and the block-diagonal A [heads * fo_pad, 2 heads] (every head's
    a1 | a2: the scores of all heads as one skinny product) -- in a handful of launches forward and backward.  Written op by op
    (a cat, and two sliced assignments per head, each with its own cast) the assembly and its autograd took ~270 launches of a
    few microseconds per step: 1.4 of the 29.8 ms of the products-sized SpGAT step, 1.9 of 12.1 ms on a rank of four.
    Inputs: W_0 .. W_{h-1} [Fin, fo], then the heads' `a` parameters (any shape with 2 fo elements, a1 first).  The gradients are
    written into the parameters' FlatAdam slots when they have them (one multi-tensor copy per parameter kind)."""

    @staticmethod
    def forward(ctx, heads, fo, fo_pad, *params):
        Ws, As = params[:heads], params[heads:]
        fin = Ws[0].shape[0]
        W = Ws[0].detach() if heads == 1 else F.cat([w.detach() for w in Ws], dim=1)
        if fo_pad != fo:
            W = F.nn.functional.pad(W.reshape(fin, heads, fo), (0, fo_pad - fo)).reshape(fin, heads * fo_pad)
        av = As[0].detach().reshape(1, -1) if heads == 1 else F.stack([a.detach().reshape(-1) for a in As])
        A = F.zeros(heads * fo_pad, 2 * heads, dtype=av.dtype, device=av.device)
        A[_pack_index(heads, fo, fo_pad, av.device)] = av.reshape(-1)
        ctx.cfg = (heads, fo, fo_pad, fin)
        ctx.targets = params
        return W, A

    @staticmethod
    def backward(ctx, gW, gA):
        from ...optim import grad_slot_of

        heads, fo, fo_pad, fin = ctx.cfg
        Ws, As = ctx.targets[:heads], ctx.targets[heads:]
        outs = [None] * (2 * heads)
        for base, grads, params in ((0, gW, Ws), (heads, gA, As)):
            if grads is None or not any(ctx.needs_input_grad[3 + base + k] for k in range(heads)):
                continue
            if base == 0:       # [heads, Fin, fo]: every head's gradient a contiguous block
                per = grads.reshape(fin, heads, fo_pad)[:, :, :fo].permute(1, 0, 2).contiguous()
            else:
                per = grads[_pack_index(heads, fo, fo_pad, grads.device)].reshape(heads, 2 * fo)
            pieces = [per[k].view(params[k].shape) for k in range(heads)]
            slots = [grad_slot_of(p) if ctx.needs_input_grad[3 + base + k] else None for k, p in enumerate(params)]
            if all(sl is not None for sl in slots):
                F._foreach_copy_(slots, pieces)           # one multi-tensor launch into the optimizer's gradient buffer
                pieces = slots
            else:
                pieces = [sl.copy_(pc) if sl is not None else pc for sl, pc in zip(slots, pieces)]
            for k in range(heads):
                if ctx.needs_input_grad[3 + base + k]:
                    outs[base + k] = pieces[k]
        return (None, None, None) + tuple(outs)


def pack_heads(Ws, a1s, a2s, fo_pad):
    """(W [Fin, heads * fo_pad], A [heads * fo_pad, 2 heads]) of a multi-head layer (see _PackHeads).  a1s / a2s: the halves of every
    head's `a` as the layers' _split_a() returns them -- views of the parameter, a1 first: the parameter itself is what enters the
    autograd node (per-view slice backward passes were most of the launches this replaces).  Any other pair is packed op by op."""
    heads, fo = len(Ws), Ws[0].shape[1]
    bases = []
    for a1, a2 in zip(a1s, a2s):
        b = a1._base
        ok = (b is not None and a2._base is b and b.numel() == 2 * fo and b.is_contiguous() and a1.stride() == (1,) and a2.stride() == (1,)
              and a1.storage_offset() == b.storage_offset() and a2.storage_offset() == b.storage_offset() + fo)
        if not ok:
            bases = None
            break
        bases.append(b)
    if bases is not None and all(W.shape == Ws[0].shape and W.dtype == Ws[0].dtype for W in Ws):
        return _PackHeads.apply(heads, fo, fo_pad, *Ws, *bases)
    if fo_pad != fo:
        Ws = [F.nn.functional.pad(W, (0, fo_pad - fo)) for W in Ws]
    W = (Ws[0] if heads == 1 else F.cat(Ws, dim=1))
    A = a1s[0].new_zeros(heads * fo_pad, 2 * heads)
    for k in range(heads):
        A[k * fo_pad:k * fo_pad + fo, k] = a1s[k]
        A[k * fo_pad:k * fo_pad + fo, heads + k] = a2s[k]
    return W, A


def _attention_dropout(graph, heads, p, training, device):
    """Per-edge, per-head multipliers of F.dropout on the attention weights (gatconv.py:37,132)."""
    if not training or p <= 0.0:
        return None
    return F.dropout(F.ones(graph.nnz, heads, device=device), p, training=True)


def _fused_heads(x, adj, Ws, a1s, a2s, alpha, concat, mode, dropout, training):
    """Shared GPU path: Ws [heads][Fin, fo], a1s/a2s [heads][fo] -> [N, heads*fo]."""
    heads, fo = len(Ws), Ws[0].shape[1]
    graph = as_csr_graph(adj)
    edge_scale = _attention_dropout(graph, heads, dropout, training, x.device)
    # sparseGatConv's form without attention dropout runs on the second-generation kernels: any per-head width that is a
    # whole number of 16-byte vectors; the max-subtracted / dropout forms need a power of two of them
    strided = mode == 0 and edge_scale is None
    fo_pad = ops.head_width_padded(fo, x.dtype, pow2=not strided)
    # The per-head padding is applied to the WEIGHTS (a [Fin, heads*fo_pad] matrix with zero columns), not to the activations:
    # the transform then writes the padded layout directly, with exact zeros, and no [N, heads*fo] matrix is copied.
    # per-node scores for every head as ONE skinny GEMM: [N, heads*fo_pad] . blockdiag(a1_k | a2_k) -> [N, 2*heads]
    # (the reference forms a[:fo].h_i + a[fo:].h_j per edge from a materialised [2*fo, E] matrix, gatconv.py:122-125)
    W, A = pack_heads(Ws, a1s, a2s, fo_pad)
    h = dense.linear(x, W)                                                     # gatconv.py:31,117 for every head at once
    A = A.to(h.dtype)
    if strided and graph.n_rows == graph.n_cols:
        # scores + aggregation as ONE autograd node (ops.gat_layer): the scores' own gradient w.r.t. h rides in the epilogue of
        # the transposed gather pass.  h is this function's own temporary (dense.linear allocates narrow rows on 128-byte
        # lines): its row padding may carry the per-node scores the gather passes fetch per edge
        out = ops.gat_layer(graph, h, A, heads, alpha, apply_elu=concat, pack_scores=True)
    else:
        st = dense.skinny_linear(h, A)                                         # fp32 [N, 2*heads]
        s, t = st[:, :heads], st[:, heads:]
        out = ops.gat_aggregate(graph, h, s, t, heads, alpha, apply_elu=concat, mode=mode, edge_scale=edge_scale,
                                pack_scores=strided)
    empty = _edgeless_rows(graph)
    if empty is not None:
        if mode == 1:
            # gatConv masks with -9e15 and soft-maxes the whole row (gatconv.py:34-36): a row without edges attends uniformly to
            # ALL nodes -- its output is the mean of Wh (through the activation); the kernels' 0 / 0 for that row is replaced
            mean = h.float().mean(0, keepdim=True).to(h.dtype)
            fill = F.elu(mean) if concat else mean
            out = out.index_copy(0, empty, fill.expand(empty.numel(), -1))
        else:
            # sparseGatConv divides 0 by 0 for such a row and asserts (gatconv.py:139-141): so does this path
            raise AssertionError("sparseGatConv: %d row(s) of the adjacency have no edge (h_prime would be NaN, gatconv.py:141)"
                                 % int(empty.numel()))
    out = _unpad_heads(out, heads, fo, fo_pad)
    if _CHECK_NAN:
        assert not F.isnan(out).any()
    return out


def _edgeless_rows(graph):
    """int64 ids of the rows without any edge, or None (the usual case); looked up once per graph object."""
    cached = getattr(graph, "_edgeless_rows", False)
    if cached is False:
        ids = F.nonzero(graph.degrees() == 0).reshape(-1)
        cached = ids if int(ids.numel()) else None
        graph._edgeless_rows = cached
    return cached


def _host_heads(x, adj, Ws, a1s, a2s, alpha, concat, mode, dropout, training):
    """The same computation for HOST tensors, on the CSR structure with torch's own segment ops (index_add /
    scatter_reduce) -- per-node scores s, t; per-edge weight from s[row] + t[col]; row-wise normalisation; weighted gather.
    No dense N x N, no [2*fo, E] edge matrix.  It exists so that the layers' host-side logic (constructors, parameter names,
    adjacency conversion) can be exercised where there is no GPU; a GPU tensor never reaches it."""
    graph = as_csr_graph(adj)
    row, col = graph.row_index(), graph.col.long()
    n = graph.n_rows
    outs = []
    for W, a1, a2 in zip(Ws, a1s, a2s):
        h = x @ W
        z = F.nn.functional.leaky_relu((h @ a1)[row] + (h @ a2)[col], alpha)
        if mode == 0:                                   # sparseGatConv: exp(-leakyrelu), gatconv.py:125
            w = F.exp(-z)
        else:                                           # gatConv: softmax of +leakyrelu over the row's edges, gatconv.py:36
            top = F.full((n,), -float("inf"), dtype=z.dtype).scatter_reduce(0, row, z, "amax")
            w = F.exp(z - top[row])
        den = F.zeros(n, dtype=h.dtype).index_add_(0, row, w)
        if training and dropout > 0.0:
            w = F.dropout(w, dropout, training=True)    # on the attention weights (gatconv.py:37,132)
        hp = F.zeros(n, h.shape[1], dtype=h.dtype).index_add_(0, row, w.unsqueeze(1) * h[col]) / den.unsqueeze(1)
        empty = den == 0
        if bool(empty.any()):
            if mode == 1:   # gatConv masks with -9e15 (gatconv.py:34-35): a row without edges attends uniformly to ALL nodes
                hp = F.where(empty.unsqueeze(1), h.mean(0, keepdim=True).expand_as(hp), hp)
            else:           # sparseGatConv divides 0 / 0 and asserts (gatconv.py:139-141)
                assert not F.isnan(hp).any()
        outs.append(F.elu(hp) if concat else hp)
    return outs[0] if len(outs) == 1 else F.cat(outs, dim=1)


def _heads(x, adj, Ws, a1s, a2s, alpha, concat, mode, dropout, training):
    fn = _fused_heads if x.is_cuda else _host_heads
    return fn(x, adj, Ws, a1s, a2s, alpha, concat, mode, dropout, training)


class gatConv(F.nn.Module):
    """Dense-adjacency GAT layer: softmax_j(leakyrelu(a1.Wh_i + a2.Wh_j)) over adj > 0 (gatconv.py:30-54)."""

    def __init__(self, in_features, out_features, dropout, alpha, concat=True):
        super().__init__()
        self.dropout, self.in_features, self.out_features = dropout, in_features, out_features
        self.alpha, self.concat = alpha, concat
        self.W = F.Parameter(F.empty(in_features, out_features))
        F.init.xavier_uniform_(self.W.data, gain=1.414)
        self.a = F.Parameter(F.empty(2 * out_features, 1))
        F.init.xavier_uniform_(self.a.data, gain=1.414)
        self.leakyrelu = F.LeakyReLU(self.alpha)

    def _split_a(self):
        return self.a[:self.out_features, 0], self.a[self.out_features:, 0]

    def forward(self, h, adj):
        a1, a2 = self._split_a()
        return _heads(h, adj, [self.W], [a1], [a2], self.alpha, self.concat, 1, self.dropout, self.training)

    def extra_repr(self):
        return "%d -> %d" % (self.in_features, self.out_features)


def _graph_of_indices(indices, shape):
    """CSRGraph for a [2, E] index tensor plus the permutation that sorts its edges row-major (None if sorted)."""
    row, col = indices[0], indices[1]
    key = row * int(shape[1]) + col
    if key.numel() > 1 and not bool((key[1:] >= key[:-1]).all()):
        order = F.argsort(key, stable=True)
        row, col = row[order], col[order]
    else:
        order = None
    return CSRGraph.from_coo(row, col, None, shape, coalesce=False), order


class SpecialSpmmFunction(F.autograd.Function):
    """sparse(indices, values, shape) @ b with gradients for `values` and `b` only (gatconv.py:60-81)."""

    @staticmethod
    def forward(ctx, indices, values, shape, b):
        assert indices.requires_grad == False  # noqa: E712  (gatconv.py:65)
        if not b.is_cuda:           # host tensors: the same three products as segment ops on the index lists
            ctx.save_for_backward(indices, values.detach(), b)
            ctx.N, ctx.gpu = int(shape[0]), False
            return F.zeros(ctx.N, b.shape[1], dtype=b.dtype).index_add_(0, indices[0], values.detach().unsqueeze(1) * b[indices[1]])
        graph, order = _graph_of_indices(indices, shape)
        vals = values.detach().to(F.float32)
        vals = vals if order is None else vals[order]
        ctx.graph, ctx.order, ctx.gpu = graph, order, True
        ctx.save_for_backward(vals, b)
        return ops.spmm_raw(graph, b, val=vals)

    @staticmethod
    def backward(ctx, grad_output):
        grad_values = grad_b = None
        if not ctx.gpu:
            indices, values, b = ctx.saved_tensors
            if ctx.needs_input_grad[1]:                                     # SDDMM: <g[row_e], b[col_e]>
                grad_values = (grad_output[indices[0]] * b[indices[1]]).sum(1)
            if ctx.needs_input_grad[3]:                                     # A^T . g
                grad_b = F.zeros_like(b).index_add_(0, indices[1], values.unsqueeze(1) * grad_output[indices[0]])
            return None, grad_values, None, grad_b
        vals, b = ctx.saved_tensors
        graph = ctx.graph
        if ctx.needs_input_grad[1]:
            grad_values = ops.sddmm_raw(graph, grad_output, b)             # gatconv.py:76-78 without the dense N x N
            if ctx.order is not None:
                unsorted = F.empty_like(grad_values)
                unsorted[ctx.order] = grad_values
                grad_values = unsorted
        if ctx.needs_input_grad[3]:
            gt, perm = graph.transpose()
            grad_b = ops.spmm_raw(gt, grad_output, val=vals[perm])         # a.t() @ grad_output, gatconv.py:80
        return None, grad_values, None, grad_b


class SpecialSpmm(F.nn.Module):
    def forward(self, indices, values, shape, b):
        return SpecialSpmmFunction.apply(indices, values, shape, b)


class sparseGatConv(F.nn.Module):
    """Sparse GAT layer: e_ij = exp(-leakyrelu(a.[h_i || h_j])), h'_i = sum_j e_ij h_j / sum_j e_ij (gatconv.py:111-148)."""

    def __init__(self, in_features, out_features, dropout, alpha, concat=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.alpha, self.concat = alpha, concat
        self.W = F.Parameter(F.zeros(in_features, out_features))
        F.init.xavier_normal_(self.W.data, gain=1.414)
        self.a = F.Parameter(F.zeros(1, 2 * out_features))
        F.init.xavier_normal_(self.a.data, gain=1.414)
        self.dropout = F.Dropout(dropout)
        self.leakyrelu = F.LeakyReLU(self.alpha)
        self.special_spmm = SpecialSpmm()

    def _split_a(self):
        return self.a[0, :self.out_features], self.a[0, self.out_features:]

    def forward(self, input, adj):
        a1, a2 = self._split_a()
        return _heads(input, adj, [self.W], [a1], [a2], self.alpha, self.concat, 0, self.dropout.p, self.training)

    def extra_repr(self):
        return "%d -> %d" % (self.in_features, self.out_features)


class _MultiHead(F.nn.Module):
    """dropout -> concat(heads) -> dropout -> elu(out head) -> log_softmax (gatconv.py:166-171, :194-199)."""
    layer_cls = None
    mode = 0

    def __init__(self, nfeat, nhid, nclass, dropout, alpha, nheads):
        super().__init__()
        self.dropout = dropout
        self.attentions = [self.layer_cls(nfeat, nhid, dropout=dropout, alpha=alpha, concat=True) for _ in range(nheads)]
        for i, attention in enumerate(self.attentions):
            self.add_module("attention_{}".format(i), attention)
        self.out_att = self.layer_cls(nhid * nheads, nclass, dropout=dropout, alpha=alpha, concat=False)

    def forward(self, x, adj):
        # bf16 activations: the log-probabilities (and the softmax of their backward) are formed in fp32 -- a bf16 log-probability
        # of -4.6 carries 1.5e-2 of absolute error, i.e. 1.5 % on the probability its backward exponentiates
        x = self.forward_activations(x, adj)
        return F.log_softmax(x, dim=1, dtype=F.float32 if x.dtype == F.bfloat16 else None)

    def forward_activations(self, x, adj):
        """forward() up to, not including, its log_softmax: elu(out head) [N, nclass].  A training loop that applies
        F.nll_loss to forward()'s output (the reference's loops) computes cross_entropy of THIS tensor; dgll_amd.ops.cross_entropy
        on it is that loss from one kernel per direction instead of log_softmax + gather and their backward passes over an fp32
        copy of the [N, nclass] output (1.4 ms of the products-sized step)."""
        x = F.dropout(x, self.dropout, training=self.training)
        first = self.attentions[0]      # every head in one launch
        halves = [att._split_a() for att in self.attentions]
        x = _heads(x, adj, [att.W for att in self.attentions], [h[0] for h in halves], [h[1] for h in halves],
                   first.alpha, True, self.mode, self.dropout, self.training)
        x = F.dropout(x, self.dropout, training=self.training)
        if x.is_cuda:       # elu(out_att(x)): the ELU (and its backward) ride in the aggregation kernel's epilogue
            out = self.out_att
            a1, a2 = out._split_a()
            x = _heads(x, adj, [out.W], [a1], [a2], out.alpha, True, self.mode,
                       out.dropout if isinstance(out.dropout, float) else out.dropout.p, self.training)
        else:
            x = F.elu(self.out_att(x, adj))
        return x


class GAT(_MultiHead):
    """Dense version of GAT (gatconv.py:154-171)."""
    layer_cls = gatConv
    mode = 1


class SpGAT(_MultiHead):
    """Sparse version of GAT (gatconv.py:174-199)."""
    layer_cls = sparseGatConv
    mode = 0
