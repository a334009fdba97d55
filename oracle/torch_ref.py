"""Restatement of the reference layers' op sequences with torch on CPU tensors (TEST INFRASTRUCTURE ONLY).

The reference's backend is torch itself (/root/reference/dgll/__init__.py:1), so the faithful CPU restatement of
its arithmetic issues the same ATen ops; autograd of these restatements gives the reference gradients at sizes the
fixtures do not cover.  Each function cites the lines it follows; each is pinned against tests/golden/*.npz in
tests/test_oracle_golden.py.  Works on CSR arrays (rowptr, col) so no dense N x N is needed.
"""
import torch


def _rows(rowptr):
    deg = rowptr[1:] - rowptr[:-1]
    return torch.repeat_interleave(torch.arange(rowptr.numel() - 1), deg)


def spmm_coo(row, col, val, x, n_rows):
    """The reference's exact call: torch.spmm(sparse_coo, dense) -- gcnconv.py:31."""
    adj = torch.sparse_coo_tensor(torch.stack([row, col]), val, (n_rows, x.shape[0]))
    return torch.spmm(adj, x)


def gcn_conv(rowptr, col, val, x, weight, bias=None):
    """gcnconv.py:29-35: support = x.W; output = spmm(adj, support) (+ bias)."""
    out = spmm_coo(_rows(rowptr), col.long(), val, torch.mm(x, weight), rowptr.numel() - 1)
    return out if bias is None else out + bias


def gcn_model(rowptr, col, val, x, w1, b1, w2, b2):
    """gcnconv.py:53-58 in eval mode (dropout inactive)."""
    h = torch.relu(gcn_conv(rowptr, col, val, x, w1, b1))
    return torch.log_softmax(gcn_conv(rowptr, col, val, h, w2, b2), dim=1)


def sage_conv(src, nbr, weight, nbr_weight, aggr="mean", hid="sum", act=True):
    """sageconv.py:32-45,70-83 with the documented fix (reduction result assigned)."""
    red = nbr.mean(1) if aggr == "mean" else nbr.sum(1) if aggr == "sum" else nbr.max(1)[0]
    nh = red @ nbr_weight
    sh = src @ weight
    out = sh + nh if hid == "sum" else torch.cat([sh, nh], 1)
    return torch.relu(out) if act else out


def sage_block(rowptr, col, x_dst, x_src, weight, nbr_weight, aggr="mean", act=True, transform_first=False, store=None):
    """sageConv on a sampled CSR block (rows = destination nodes, columns index x_src): the K-axis reduce of
    sageconv.py:33-36 over a ragged neighbour list (a seed with fewer than `fanout` neighbours keeps them all,
    base_sampler.py:53-54), then sageconv.py:41,72-75.  transform_first: reduce(x_src.W_n) instead of reduce(x_src).W_n
    (equal in exact arithmetic; the build aggregates the narrow product when a layer narrows).  `store`: optional
    rounding applied where the GPU path stores a tensor (bf16 emulation with a straight-through gradient)."""
    store = store or (lambda t: t)
    n = rowptr.numel() - 1
    row, colv = _rows(rowptr), col.long()
    deg = (rowptr[1:] - rowptr[:-1]).clamp(min=1).to(x_src.dtype)

    def reduce(m):
        # sum, THEN scale (torch.mean's order, sageconv.py:33-36; oracle.c:70 the same): scaling every term first is not
        # correctly rounded where a mean of bf16-quantised rows lands on a bf16 rounding tie, which is not rare
        total = torch.zeros(n, m.shape[1], dtype=m.dtype).index_add_(0, row, m[colv])
        return total * (1.0 / deg)[:, None] if aggr == "mean" else total

    if transform_first:
        nh = store(reduce(store(x_src @ nbr_weight)))
    else:
        nh = store(reduce(x_src)) @ nbr_weight
    out = x_dst @ weight + nh
    return store(torch.relu(out) if act else out)


def spgat_conv(rowptr, col, x, W, a, alpha, concat=True, heads=1):
    """sparseGatConv.forward, gatconv.py:111-148 (dropout inactive); multi-head when W is [heads, Fin, Fo] and a is
    [heads, 1, 2*Fo] (the torch.cat over heads of SpGAT.forward, gatconv.py:196)."""
    if W.dim() == 2:
        W, a = W[None], a[None]
    row, colv = _rows(rowptr), col.long()
    n = rowptr.numel() - 1
    outs = []
    for k in range(W.shape[0]):
        fo = W.shape[2]
        h = x @ W[k]
        z = (h @ a[k, 0, :fo])[row] + (h @ a[k, 0, fo:])[colv]
        e = torch.exp(-torch.nn.functional.leaky_relu(z, alpha))
        den = torch.zeros(n, dtype=h.dtype).index_add_(0, row, e)
        num = torch.zeros(n, fo, dtype=h.dtype).index_add_(0, row, e[:, None] * h[colv])
        hp = num / den[:, None]
        outs.append(torch.nn.functional.elu(hp) if concat else hp)
    return torch.cat(outs, 1)


def gat_conv(rowptr, col, x, W, a, alpha, concat=True):
    """gatConv.forward restricted to the adjacency's nonzeros, gatconv.py:30-54: softmax_j(+leakyrelu) with the
    row maximum subtracted; valid when every row has an edge.  W [heads, Fin, Fo] / a [heads, 2*Fo, 1] for heads."""
    if W.dim() == 2:
        W, a = W[None], a[None]
    row, colv = _rows(rowptr), col.long()
    n = rowptr.numel() - 1
    outs = []
    for k in range(W.shape[0]):
        fo = W.shape[2]
        h = x @ W[k]
        z = torch.nn.functional.leaky_relu((h @ a[k, :fo, 0])[row] + (h @ a[k, fo:, 0])[colv], alpha)
        mx = torch.full((n,), -float("inf"), dtype=h.dtype).scatter_reduce(0, row, z, "amax")
        e = torch.exp(z - mx[row])
        den = torch.zeros(n, dtype=h.dtype).index_add_(0, row, e)
        hp = torch.zeros(n, fo, dtype=h.dtype).index_add_(0, row, (e / den[row])[:, None] * h[colv])
        outs.append(torch.nn.functional.elu(hp) if concat else hp)
    return torch.cat(outs, 1)


def multihead_model(kind, rowptr, col, x, W, a, W_out, a_out, alpha):
    """GAT / SpGAT forward in eval mode, gatconv.py:166-171 / :194-199."""
    conv = spgat_conv if kind == "spgat" else gat_conv
    h = conv(rowptr, col, x, W, a, alpha, True)
    o = torch.nn.functional.elu(conv(rowptr, col, h, W_out, a_out, alpha, False))
    return torch.log_softmax(o, dim=1)


def special_spmm(rowptr, col, values, b):
    """SpecialSpmmFunction.forward, gatconv.py:66-69 (autograd gives the :72-81 gradients)."""
    return spmm_coo(_rows(rowptr), col.long(), values, b, rowptr.numel() - 1)


def scatter_pool(x, batch, size=None, reduce="sum"):
    """Graph read-out the reference gets from torch_scatter.scatter(x, batch, dim=0, dim_size=size, reduce=...)
    (dgll/nn/GlobalPooling/Pooling.py:37,59,81).  torch_scatter is an un-vendored, unpinned dependency
    (requirements.txt:2) that is not installed here, so this restates its published semantics -- 'add'/'sum': segment
    sum; 'mean': sum / max(count, 1); 'max': segment maximum, 0 for segments without entries -- and is PARITY UNPINNED:
    no golden vector of the reference exists for it."""
    size = int(batch.max()) + 1 if size is None else size
    idx = batch.long().view(-1, 1).expand(-1, x.shape[1])
    if reduce in ("sum", "add", "mean"):
        out = torch.zeros(size, x.shape[1], dtype=x.dtype).scatter_add_(0, idx, x)
        if reduce == "mean":
            cnt = torch.bincount(batch.long(), minlength=size).clamp(min=1).to(x.dtype)
            out = out / cnt.unsqueeze(1)
        return out
    if reduce == "max":
        # torch_scatter's max returns ONE arg-max per (segment, feature) and routes the gradient there; ties go to
        # the first entry in node order (its CPU loop updates on strict '>').
        n = x.shape[0]
        m = torch.full((size, x.shape[1]), float("-inf"), dtype=x.dtype).scatter_reduce_(0, idx, x.detach(), "amax", include_self=True)
        pos = torch.arange(n).view(-1, 1).expand(-1, x.shape[1])
        cand = torch.where(x.detach() == m[batch.long()], pos, torch.full_like(pos, n))
        arg = torch.full((size, x.shape[1]), n, dtype=torch.int64).scatter_reduce_(0, idx, cand, "amin", include_self=True)
        has = arg < n
        return torch.where(has, x.gather(0, arg.clamp(max=max(n - 1, 0))), torch.zeros((), dtype=x.dtype))
    raise ValueError(reduce)
