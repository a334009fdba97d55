"""Adjacency preparation on the device (row f4 / a11): what the reference does on the host with scipy before handing
the layers a torch COO tensor (/root/reference/dgll/nn/utils/utils.py:163-171, 240-257; dgll/nn/utililities.py:23-47):

    adj = adj + adj.T.multiply(adj.T > adj) - adj.multiply(adj.T > adj)     # symmetrise          utils.py:164
    adj = normalize(adj + sp.eye(n))                                         # D^-1 (A + I)        utils.py:171,240-247
    adj = sparse_mx_to_torch_sparse_tensor(adj)                              # fp32 COO, sorted    utils.py:250-257

Here the same three steps act on edge lists / CSRGraph with torch ops on whatever device the edges live on, and return
the CSR the kernels consume (plus `to_torch_coo` for callers that want the reference's tensor type back)."""
import torch

from .graph import CSRGraph


def symmetrize(row, col, val=None):
    """max(A, A^T) on the union pattern -- utils.py:164 for non-negative weights."""
    r = torch.cat([row, col])
    c = torch.cat([col, row])
    v = None if val is None else torch.cat([val, val])
    return r, c, v


def normalized_adjacency(row, col, n, val=None, symmetric=True, self_loops=True):
    """D^-1 (sym(A) + I) as a CSRGraph with fp32 values (rows summing to 0 keep zeros, utils.py:243-244)."""
    row, col = row.to(torch.int64), col.to(torch.int64)
    val = torch.ones(row.numel(), dtype=torch.float32, device=row.device) if val is None else val.to(torch.float32)
    if symmetric:
        # coalesce first so that symmetrising takes max(a_ij, a_ji) rather than a sum
        key = torch.cat([row * n + col, col * n + row])
        v = torch.cat([val, val])
        uniq, inv = torch.unique(key, return_inverse=True)
        vmax = torch.zeros(uniq.numel(), dtype=torch.float32, device=row.device).scatter_reduce(0, inv, v, "amax", include_self=False)
        row, col, val = torch.div(uniq, n, rounding_mode="floor"), uniq % n, vmax
    if self_loops:
        eye = torch.arange(n, dtype=torch.int64, device=row.device)
        row, col = torch.cat([row, eye]), torch.cat([col, eye])
        val = torch.cat([val, torch.ones(n, dtype=torch.float32, device=row.device)])
    g = CSRGraph.from_coo(row, col, val, (n, n))          # sorts, sums duplicates (A + I where A already had a loop)
    rowsum = torch.zeros(n, dtype=torch.float32, device=g.device).index_add_(0, g.row_index(), g.val)
    r_inv = torch.where(rowsum != 0, 1.0 / rowsum, torch.zeros_like(rowsum))
    g.val = g.val * r_inv[g.row_index()]
    return g


def to_torch_coo(graph):
    """The reference's hand-over type: sorted fp32 COO (utils.py:250-257)."""
    ind = torch.stack([graph.row_index(), graph.col.to(torch.int64)])
    val = graph.val if graph.val is not None else torch.ones(graph.nnz, dtype=torch.float32, device=graph.device)
    return torch.sparse_coo_tensor(ind, val, graph.shape)
