"""Python face of the reference's FusedKernel extension (`gcn_extension`, /root/reference/dgll/FusedKernel/
gcn_extension.cpp:103-110) and of the classes its training script builds on it (train_gcn.py:8-57).

    gcn_fused_forward(row_ptr, col_idx, values, X, W, num_neighbors, actual_F) -> H        gcn_extension.cpp:22-58
    gcn_fused_backward(grad_output, row_ptr, col_idx, values, X, W, num_neighbors, actual_F) -> [grad_X, grad_W]   :60-101
    GCNFusedFunction / GCNLayer / GCN                                                       train_gcn.py:8-57

The forward calls the reference-named C symbol's twin (dgll_hip_gcn_fused_forward) in libdgll_hip.so.  The backward
follows the MATH of relu(A.(X.W)) -- the reference's backward kernel ignores the ReLU mask, writes grad_X to the wrong
row and races on shared memory, which is not reproduced.
"""
import torch

from . import _lib, dense, ops
from .graph import CSRGraph


def _check(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("%s must be a CUDA tensor" % "argument")   # TORCH_CHECK(..is_cuda()), gcn_extension.cpp:31-36


def gcn_fused_forward(row_ptr, col_idx, values, X, W, num_neighbors, actual_F):
    _check(row_ptr, col_idx, values, X, W, num_neighbors)
    N, F_padded = X.shape
    H_dim = W.shape[1]
    row_ptr = row_ptr.to(torch.int32).contiguous()
    col_idx = col_idx.to(torch.int32).contiguous()
    values, X, W = values.float().contiguous(), X.float().contiguous(), W.float().contiguous()
    H = torch.zeros((N, H_dim), dtype=torch.float32, device=X.device)            # gcn_extension.cpp:43-44
    ws_bytes = int(_lib.lib.dgll_hip_gcn_fused_workspace_bytes(N, int(actual_F), H_dim))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=X.device)
    with torch.cuda.device(X.device):
        code = _lib.lib.dgll_hip_gcn_fused_forward(
            torch.cuda.current_stream(X.device).cuda_stream, row_ptr.data_ptr(), col_idx.data_ptr(), values.data_ptr(),
            X.data_ptr(), W.data_ptr(), H.data_ptr(), N, F_padded, int(actual_F), H_dim, int(col_idx.numel()),
            ws.data_ptr(), ws_bytes)
    _lib.check(code, "dgll_hip_gcn_fused_forward")
    return H


_graphs = {}


def _graph(row_ptr, col_idx, values):
    key = (row_ptr.data_ptr(), col_idx.data_ptr(), values.data_ptr(), values._version)
    g = _graphs.get(key)
    if g is None:
        if len(_graphs) > 16:
            _graphs.clear()
        n = row_ptr.numel() - 1
        g = _graphs[key] = CSRGraph(row_ptr.to(torch.int64), col_idx.to(torch.int32), values.float(), n, n)
    return g


def gcn_fused_backward(grad_output, row_ptr, col_idx, values, X, W, num_neighbors, actual_F):
    _check(grad_output, row_ptr, col_idx, values, X, W, num_neighbors)
    F = int(actual_F)
    g = _graph(row_ptr, col_idx, values)
    Xa, Wa = X[:, :F].float(), W[:F].float()
    AX = ops.spmm_raw(g, Xa)
    mask = dense.mm_nt(AX, Wa.t()) > 0                           # the ReLU of the forward (gcn_fused_kernel.cu:66)
    G = grad_output.float() * mask
    grad_W = torch.zeros_like(W)
    grad_W[:F] = dense.grad_weight(AX, G)
    grad_X = torch.zeros_like(X)
    gt, _ = g.transpose()
    grad_X[:, :F] = ops.spmm_raw(gt, dense.mm_nt(G, Wa))
    return [grad_X, grad_W]


class GCNFusedFunction(torch.autograd.Function):
    """train_gcn.py:8-22."""

    @staticmethod
    def forward(ctx, row_ptr, col_idx, values, X, W, num_neighbors, actual_F):
        output = gcn_fused_forward(row_ptr, col_idx, values, X, W, num_neighbors, actual_F)
        ctx.save_for_backward(row_ptr, col_idx, values, X, W, num_neighbors)
        ctx.actual_F = actual_F
        return output

    @staticmethod
    def backward(ctx, grad_output):
        row_ptr, col_idx, values, X, W, num_neighbors = ctx.saved_tensors
        grad_X, grad_W = gcn_fused_backward(grad_output, row_ptr, col_idx, values, X, W, num_neighbors, ctx.actual_F)
        return None, None, None, grad_X, grad_W, None, None


class GCNLayer(torch.nn.Module):
    """train_gcn.py:24-42: W [in_features_padded, out_features] ~ randn / sqrt(actual_in_features)."""

    def __init__(self, in_features_padded, actual_in_features, out_features, device="cuda"):
        super().__init__()
        scale = 1.0 / float(actual_in_features) ** 0.5
        self.W = torch.nn.Parameter(torch.randn(in_features_padded, out_features, dtype=torch.float32, device=device) * scale)
        self.actual_F = actual_in_features

    def forward(self, row_ptr, col_idx, values, X, num_neighbors):
        return GCNFusedFunction.apply(row_ptr, col_idx, values, X, self.W, num_neighbors, self.actual_F)


class GCN(torch.nn.Module):
    """train_gcn.py:44-57: two fused layers, input features zero-padded to a multiple of 4."""

    def __init__(self, input_dim, hidden_dim, output_dim, device="cuda"):
        super().__init__()
        self.input_dim_padded = ((input_dim + 3) // 4) * 4
        self.layer1 = GCNLayer(self.input_dim_padded, input_dim, hidden_dim, device)
        self.layer2 = GCNLayer(hidden_dim, hidden_dim, output_dim, device)

    def forward(self, row_ptr, col_idx, values, X, num_neighbors):
        X_padded = torch.zeros(X.size(0), self.input_dim_padded, device=X.device, dtype=torch.float32)
        X_padded[:, :X.size(1)] = X
        h1 = self.layer1(row_ptr, col_idx, values, X_padded, num_neighbors)
        return self.layer2(row_ptr, col_idx, values, h1, num_neighbors)


def csr_from_edge_index(edge_index, num_nodes):
    """Symmetric-normalised D^-1/2 A D^-1/2 in int32 CSR as train_gcn.py:60-78 prepares it -- with row_ptr built from
    nnz COUNTS (the reference cumsums the row VALUE sums, train_gcn.py:74-75, which is only right for unit weights)."""
    row, col = edge_index[0], edge_index[1]
    ones = torch.ones(row.numel(), dtype=torch.float32, device=row.device)
    deg = torch.zeros(num_nodes, dtype=torch.float32, device=row.device).index_add_(0, row, ones)
    dinv = deg.pow(-0.5)
    dinv[deg == 0] = 0
    g = CSRGraph.from_coo(row, col, dinv[row] * dinv[col], (num_nodes, num_nodes))
    return g.rowptr.to(torch.int32), g.col, g.val, g.degrees().to(torch.int32)
