"""GCN layer on the gfx950 aggregation engine.

Interface mirror of /root/reference/dgll/nn/Convolution/gcnconv.py: `gcnConv(in_features, out_features,
bias=True)` with parameters `weight` [in, out] and `bias` [out] initialised U(-1/sqrt(out), 1/sqrt(out))
(gcnconv.py:23-27), `forward(x, adj)` = adj . (x . weight) + bias (gcnconv.py:29-35), and the two-layer
`GCN(in_features, nhid, nclass, dropout)` example (gcnconv.py:43-58).  state_dicts interchange.

On GPU tensors the product with `adj` runs in dgll_hip_spmm_csr with the bias add fused into the kernel's
epilogue; `adj` may be the reference's torch sparse COO tensor (converted to CSR once and cached) or a
dgll_amd.CSRGraph.
"""
import math

from ... import backend as F
from ... import dense, ops
from ...graph import as_csr_graph


class gcnConv(F.nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = F.Parameter(F.empty(in_features, out_features))
        if bias:
            self.bias = F.Parameter(F.empty(out_features))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1.0 / math.sqrt(self.out_features)  # gcnconv.py:24: 1/sqrt(weight.size(1))
        with F.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def forward(self, x, adj, relu=False):
        """adj . (x . weight) + bias (gcnconv.py:29-35).  relu=True (an extension GCN.forward uses) applies the activation
        that always follows inside the aggregation kernel's epilogue instead of as a pass of its own."""
        if x.is_cuda:
            support = dense.linear(x, self.weight)           # transform first (gcnconv.py:30): MFMA kernel for bf16 inputs
            return ops.spmm(as_csr_graph(adj), support, bias=self.bias, relu=relu)   # aggregate + fused bias (:31-33)
        support = F.mm(x, self.weight.to(x.dtype))
        out = F.spmm(adj, support)
        out = out if self.bias is None else out + self.bias
        return F.relu(out) if relu else out

    def extra_repr(self):
        return "%d -> %d" % (self.in_features, self.out_features)


class GCN(F.nn.Module):
    """log_softmax(gcn2(dropout(relu(gcn1(x, A))), A)) -- gcnconv.py:53-58."""

    def __init__(self, in_features, nhid, nclass, dropout):
        super().__init__()
        self.in_features, self.nhid, self.nclass, self.dropout = in_features, nhid, nclass, dropout
        self.gcn1 = gcnConv(in_features, nhid)
        self.gcn2 = gcnConv(nhid, nclass)

    def forward(self, x, adj):
        hidden = self.gcn1(x, adj, relu=True)                 # F.relu(self.gcn1(x, adj)), gcnconv.py:54
        hidden = F.dropout(hidden, self.dropout, training=self.training)
        out = self.gcn2(hidden, adj)
        return F.log_softmax(out, dim=1, dtype=F.float32 if out.dtype == F.bfloat16 else None)   # fp32 log-probabilities (gatconv.py note)
