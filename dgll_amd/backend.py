"""dgll.backend -- the `F` object every reference module imports (`from dgll import backend as F`).

In the reference this is literally torch (/root/reference/dgll/__init__.py:1: `import torch as backend`), and
the layers additionally use six names torch does not have (F.Parameter, F.init, F.LeakyReLU, F.Dropout, F.elu,
F.dropout(..., training=) -- gcnconv.py:15,55, sageconv.py:20,28, gatconv.py:23-28,41,107-108).  This module
forwards everything to torch, supplies those names, and overrides the sparse aggregation entry points so that
an adjacency on the GPU is multiplied by the hand-written gfx950 kernels:

    F.spmm(adj, dense)          gcnconv.py:31, gcn.py:39
    F.sparse.mm(adj, dense)     Evaluation/PPI/gcn_model.py:76 (via torch.sparse.mm)
    F.mm / F.matmul (2-D)       gcnconv.py:30, sageconv.py:41,72, gatconv.py:31,117 -> the MFMA / fp32 matrix-core kernels

CPU tensors take torch's own CPU op, exactly as they do in the reference (its plumbing configuration, BASELINE
config 1, is CPU-only); a GPU tensor never falls back to anything.
"""
import sys
import types

import torch

from . import ops
from .graph import CSRGraph, as_csr_graph

# ---- names the reference expects on F but torch lacks ----------------------------------------------------
Parameter = torch.nn.Parameter
init = torch.nn.init
LeakyReLU = torch.nn.LeakyReLU
Dropout = torch.nn.Dropout
elu = torch.nn.functional.elu
dropout = torch.nn.functional.dropout
relu = torch.nn.functional.relu
log_softmax = torch.nn.functional.log_softmax


def _on_gpu(adj, dense):
    a_cuda = adj.is_cuda
    if a_cuda != dense.is_cuda:
        raise RuntimeError("adjacency and features must live on the same device")
    return a_cuda


def spmm(adj, dense):
    """A.X with A sparse (torch COO/CSR tensor or CSRGraph) -- gcnconv.py:31."""
    if _on_gpu(adj, dense):
        return ops.spmm(as_csr_graph(adj), dense)
    if isinstance(adj, CSRGraph):
        adj = torch.sparse_csr_tensor(adj.rowptr, adj.col.long(),
                                      adj.val if adj.val is not None else torch.ones(adj.nnz), adj.shape)
    return torch.spmm(adj, dense)


def mm(a, b):
    """x . W (gcnconv.py:30, gatconv.py:31,117): GPU matrices run on the hand-written dense kernels, with autograd."""
    if a.is_cuda and a.dim() == 2 and b.dim() == 2:
        from . import dense

        return dense.linear(a, b)
    return torch.mm(a, b)


def matmul(a, b):
    """F.matmul of sageconv.py:41,72: as `mm` for two GPU matrices, torch.matmul for everything else (host, batched)."""
    if a.is_cuda and a.dim() == 2 and b.dim() == 2:
        from . import dense

        return dense.linear(a, b)
    return torch.matmul(a, b)


class _Sparse(types.ModuleType):
    """F.sparse: torch.sparse with mm() routed like F.spmm."""

    def __getattr__(self, name):
        return getattr(torch.sparse, name)

    @staticmethod
    def mm(adj, dense):
        return spmm(adj, dense)


sparse = _Sparse("dgll_amd.backend.sparse")


def __getattr__(name):  # everything else is torch's (PEP 562)
    return getattr(torch, name)


sys.modules.setdefault(__name__ + ".sparse", sparse)
