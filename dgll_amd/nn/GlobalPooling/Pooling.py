"""Global (graph-level) pooling: sum / mean / max of node rows per graph of a batch.

Mirrors dgll/nn/GlobalPooling/Pooling.py:18-119 (sumPooling :18-37, meanPooling :40-59, maxPooling :62-81,
class Pooling :83-119).  The reference delegates to `torch_scatter.scatter(x, batch, dim=0, dim_size=size,
reduce=...)` (a dependency that is not installed here); its published semantics are restated in
oracle/torch_ref.py:scatter_pool -- segments with no node give 0 for every reduce.

Here a batch vector becomes a CSR "segment matrix" (row g lists the node rows of graph g) once, and the reductions are
the engine's own kernels: the CSR SpMM (sum / mean; graphs of thousands of nodes take its chunked long-row path) and
segment-max (gradient to the arg-max row, as torch_scatter's max).  No atomics: results are run-to-run identical.
"""
from typing import List, Optional, Union

import torch

from ... import ops
from ...graph import CSRGraph


def segments_of(batch, size=None):
    """CSRGraph [B, N] whose row g gathers the nodes with batch == g (any order of `batch`; stable within a graph)."""
    if batch.dim() != 1:
        raise ValueError("batch must be a 1-D vector of graph ids")
    n = batch.numel()
    if n and (int(batch.min()) < 0):
        raise ValueError("batch ids must be >= 0")
    size = (int(batch.max().item()) + 1 if n else 0) if size is None else int(size)
    if n and int(batch.max()) >= size:
        raise ValueError("batch id %d outside size %d" % (int(batch.max()), size))
    b = batch.to(torch.int64)
    rowptr = torch.zeros(size + 1, dtype=torch.int64, device=batch.device)
    if n:
        rowptr[1:] = torch.cumsum(torch.bincount(b, minlength=size), 0)
    sorted_already = n < 2 or bool((b[1:] >= b[:-1]).all())
    col = torch.arange(n, dtype=torch.int32, device=batch.device) if sorted_already else torch.sort(b, stable=True)[1].to(torch.int32)
    return CSRGraph(rowptr, col, None, size, n, check=False)


def _segments(x, batch, size):
    if not x.is_cuda:
        raise RuntimeError("dgll_amd pooling runs on the GPU (HIP kernels); got a %s tensor" % x.device.type)
    if batch.numel() != x.shape[0]:
        raise ValueError("batch has %d entries for %d node rows" % (batch.numel(), x.shape[0]))
    return segments_of(batch.to(x.device), size)


def sumPooling(x: torch.Tensor, batch: Optional[torch.Tensor], size: Optional[int] = None) -> torch.Tensor:
    """r_g = sum of the node rows of graph g (Pooling.py:18-37)."""
    if batch is None:
        return x.sum(dim=0, keepdim=True)
    return ops.spmm(_segments(x, batch, size), x, reduce="sum")


def meanPooling(x: torch.Tensor, batch: Optional[torch.Tensor], size: Optional[int] = None) -> torch.Tensor:
    """r_g = mean of the node rows of graph g (Pooling.py:40-59)."""
    if batch is None:
        return x.mean(dim=0, keepdim=True)
    return ops.spmm(_segments(x, batch, size), x, reduce="mean")


def maxPooling(x: torch.Tensor, batch: Optional[torch.Tensor], size: Optional[int] = None) -> torch.Tensor:
    """r_g = feature-wise max over the node rows of graph g (Pooling.py:62-81)."""
    if batch is None:
        return x.max(dim=0, keepdim=True)[0]
    return ops.segment_max(_segments(x, batch, size), x)


class Pooling(torch.nn.Module):
    """Wrapper choosing one or several read-outs, concatenated on the last dimension (Pooling.py:83-119)."""

    def __init__(self, aggr: Union[str, List[str]]):
        super().__init__()
        self.aggrs = [aggr] if isinstance(aggr, str) else list(aggr)
        if not self.aggrs or any(a not in ("sum", "add", "mean", "max") for a in self.aggrs):
            raise ValueError("aggr must be one or more of 'sum', 'add', 'mean', 'max'")

    def forward(self, x, batch, size: Optional[int] = None):
        if batch is None:
            seg = None
        else:
            seg = _segments(x, batch, size)   # one segment matrix (and its launch plan) for all read-outs
        xs = []
        for aggr in self.aggrs:
            if seg is None:
                xs.append({"sum": sumPooling, "add": sumPooling, "mean": meanPooling, "max": maxPooling}[aggr](x, None))
            elif aggr == "max":
                xs.append(ops.segment_max(seg, x))
            else:
                xs.append(ops.spmm(seg, x, reduce="mean" if aggr == "mean" else "sum"))
        return xs[0] if len(xs) == 1 else torch.cat(xs, dim=-1)

    def __repr__(self):
        aggr = self.aggrs[0] if len(self.aggrs) == 1 else self.aggrs
        return "%s(aggr=%s)" % (self.__class__.__name__, aggr)
