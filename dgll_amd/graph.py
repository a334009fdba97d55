"""CSRGraph -- the adjacency container the HIP kernels consume.

The reference hands its layers a torch sparse COO tensor with sorted, unique int64 indices and fp32 values
(dgll/nn/utils/utils.py:250-257), a dense 0/1 matrix whose nonzeros are taken row-major
(dgll/nn/Convolution/gatconv.py:115), an un-coalesced all-ones COO rebuilt from `edge_index`
(Evaluation/PPI/gcn_model.py:56), or a dense [N, K, D] neighbour tensor (sageconv.py:70).  All of them
become one structure here: int64 rowptr, int32 col, optional fp32 values, resident on the device, built
once and cached together with its load-balancing plan and (lazily) its transpose for the backward pass.
"""
import ctypes as C
import weakref

import torch

from . import _lib


class CSRGraph:
    """CSR adjacency (`n_rows` destination rows gathering from `n_cols` source rows) on one device."""

    def __init__(self, rowptr, col, val=None, n_rows=None, n_cols=None, check=True):
        n_rows = int(rowptr.numel() - 1) if n_rows is None else int(n_rows)
        if check:
            if rowptr.dtype != torch.int64 or col.dtype != torch.int32:
                raise TypeError("CSRGraph needs int64 rowptr and int32 col")
            if rowptr.dim() != 1 or rowptr.numel() != n_rows + 1:
                raise ValueError("rowptr must have n_rows + 1 entries")
            if val is not None and (val.dtype != torch.float32 or val.numel() != col.numel()):
                raise ValueError("val must be fp32 with one entry per nonzero")
        self.rowptr = rowptr.contiguous()
        self.col = col.contiguous()
        self.val = None if val is None else val.contiguous()
        self.n_rows = n_rows
        self.n_cols = int(n_cols) if n_cols is not None else (int(col.max()) + 1 if col.numel() else 0)
        self.nnz = int(col.numel())
        self._plan = None
        self._plan_finalizer = None
        self._transpose = None  # (CSRGraph of A^T, perm) with A^T.edge[k] == A.edge[perm[k]]
        self._deg = None

    # ------------------------------------------------------------------ properties
    @property
    def device(self):
        return self.rowptr.device

    @property
    def is_cuda(self):
        return self.rowptr.is_cuda

    @property
    def shape(self):
        return (self.n_rows, self.n_cols)

    def __repr__(self):
        return "CSRGraph(n_rows=%d, n_cols=%d, nnz=%d, weighted=%s, device=%s)" % (
            self.n_rows, self.n_cols, self.nnz, self.val is not None, self.device)

    def degrees(self):
        if self._deg is None:
            self._deg = self.rowptr[1:] - self.rowptr[:-1]
        return self._deg

    max_degree = None        # upper bound on the row length when the constructor knows one (sampled blocks: the fan-out)
    identity_cols = False    # True for sampled blocks: col == arange(nnz), every source row belongs to exactly one edge

    def row_index(self):
        """Expanded int64 row id of every nonzero (COO row vector)."""
        if self.identity_cols:           # blocks are rebuilt per mini-batch and need it in the backward pass: keep it
            if getattr(self, "_row_index", None) is None:
                self._row_index = torch.repeat_interleave(torch.arange(self.n_rows, device=self.device), self.degrees(),
                                                          output_size=self.nnz)
            return self._row_index
        return torch.repeat_interleave(torch.arange(self.n_rows, device=self.device), self.degrees())

    def with_values(self, val):
        """Same structure (shared plan/transpose caches), different edge values."""
        g = CSRGraph(self.rowptr, self.col, val, self.n_rows, self.n_cols, check=False)
        g._plan, g._transpose, g._deg = self.plan() if self.is_cuda else None, self._transpose, self._deg
        g._structure_owner = self  # keep the plan's owner alive
        return g

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_coo(cls, row, col, val=None, shape=None, coalesce=True):
        """Build from COO triples on any device.  Duplicates are summed (torch coalesce semantics, which is
        what torch.sparse.mm applies implicitly to Evaluation/PPI/gcn_model.py:56's un-coalesced tensor)."""
        row = row.to(torch.int64)
        col = col.to(torch.int64)
        if shape is None:
            shape = (int(row.max()) + 1 if row.numel() else 0, int(col.max()) + 1 if col.numel() else 0)
        n_rows, n_cols = int(shape[0]), int(shape[1])
        if n_cols >= 2 ** 31:
            raise ValueError("column ids must fit int32")
        if row.numel():
            key = row * n_cols + col
            sorted_already = bool((key[1:] > key[:-1]).all()) if key.numel() > 1 else True
            if not sorted_already:
                key, order = torch.sort(key, stable=True)
                if val is not None:
                    val = val[order]
                if coalesce:
                    uniq, inverse = torch.unique_consecutive(key, return_inverse=True)
                    if uniq.numel() != key.numel():
                        summed = torch.zeros(uniq.numel(), dtype=torch.float32, device=key.device)
                        ones = val if val is not None else torch.ones(key.numel(), dtype=torch.float32, device=key.device)
                        summed.index_add_(0, inverse, ones.to(torch.float32))
                        key, val = uniq, summed
                row = torch.div(key, n_cols, rounding_mode="floor")
                col = key - row * n_cols
        counts = torch.bincount(row, minlength=n_rows) if row.numel() else torch.zeros(n_rows, dtype=torch.int64, device=row.device)
        rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=row.device)
        torch.cumsum(counts, 0, out=rowptr[1:])
        return cls(rowptr, col.to(torch.int32), None if val is None else val.to(torch.float32), n_rows, n_cols)

    @classmethod
    def from_torch_sparse(cls, adj):
        """torch sparse COO / CSR tensor -> CSRGraph (values kept as given, fp32)."""
        if adj.layout == torch.sparse_csr:
            return cls(adj.crow_indices().to(torch.int64), adj.col_indices().to(torch.int32),
                       adj.values().to(torch.float32), adj.shape[0], adj.shape[1])
        if adj.layout != torch.sparse_coo:
            raise TypeError("expected a torch sparse COO or CSR tensor")
        ind = adj._indices()
        return cls.from_coo(ind[0], ind[1], adj._values(), adj.shape)

    @classmethod
    def from_dense(cls, adj, weighted=False):
        """Nonzero pattern of a dense matrix in row-major order -- `adj.nonzero().t()` of gatconv.py:115."""
        edge = adj.nonzero()
        val = adj[edge[:, 0], edge[:, 1]].to(torch.float32) if weighted else None
        return cls.from_coo(edge[:, 0], edge[:, 1], val, adj.shape)

    @classmethod
    def from_edge_index(cls, edge_index, num_nodes):
        """All-ones adjacency from a [2, E] index tensor with duplicates summed (PPI/gcn_model.py:44-57)."""
        ones = torch.ones(edge_index.shape[1], dtype=torch.float32, device=edge_index.device)
        return cls.from_coo(edge_index[0], edge_index[1], ones, (num_nodes, num_nodes))

    @classmethod
    def fixed_fanout(cls, n, k, device):
        """Row i gathers rows i*k .. i*k+k-1: the [N, K, D] neighbour tensor of sageconv.py:70 viewed as
        [N*K, D] (rowptr = arange(0, N*K+1, K), col = arange(N*K))."""
        if n * k >= 2 ** 31:
            raise ValueError("N*K must fit int32")
        rowptr = torch.arange(0, n * k + 1, k, dtype=torch.int64, device=device) if k > 0 else torch.zeros(n + 1, dtype=torch.int64, device=device)
        col = torch.arange(n * k, dtype=torch.int32, device=device)
        g = cls(rowptr, col, None, n, n * k, check=False)
        g.identity_cols = True
        g.max_degree = k
        return g

    def to(self, device):
        device = torch.device(device)
        if device == self.device:
            return self
        return CSRGraph(self.rowptr.to(device), self.col.to(device), None if self.val is None else self.val.to(device),
                        self.n_rows, self.n_cols, check=False)

    # ------------------------------------------------------------------ transpose (backward pass)
    def transpose(self):
        """(A^T as CSRGraph, perm): built once with a stable sort, so every reduction order is fixed and the
        backward pass needs no atomics.  perm maps A^T's edge slots to A's (for per-edge values)."""
        if self._transpose is None:
            if self.nnz:
                key = self.col.to(torch.int64) * self.n_rows + self.row_index()
                key, perm = torch.sort(key, stable=True)
                t_row = torch.div(key, self.n_rows, rounding_mode="floor")
                t_col = (key - t_row * self.n_rows).to(torch.int32)
                counts = torch.bincount(t_row, minlength=self.n_cols)
            else:
                perm = torch.zeros(0, dtype=torch.int64, device=self.device)
                t_col = torch.zeros(0, dtype=torch.int32, device=self.device)
                counts = torch.zeros(self.n_cols, dtype=torch.int64, device=self.device)
            rowptr = torch.zeros(self.n_cols + 1, dtype=torch.int64, device=self.device)
            torch.cumsum(counts, 0, out=rowptr[1:])
            gt = CSRGraph(rowptr, t_col, None if self.val is None else self.val[perm], self.n_cols, self.n_rows, check=False)
            self._transpose = (gt, perm)
        return self._transpose

    def inv_degrees(self):
        """fp32[n_rows]: 1 / max(deg, 1) -- the mean reduce's row factor; cached."""
        if getattr(self, "_inv_deg", None) is None:
            self._inv_deg = (1.0 / self.degrees().clamp(min=1).to(torch.float32)).contiguous()
        return self._inv_deg

    def mean_scale_transposed(self):
        """fp32[nnz] in A^T edge order: 1/deg(i) of the destination row i each edge came from (backward of the
        mean reduce); cached."""
        if getattr(self, "_mean_scale_t", None) is None:
            gt, _ = self.transpose()
            inv = 1.0 / self.degrees().clamp(min=1).to(torch.float32)
            self._mean_scale_t = inv[gt.col.long()]
        return self._mean_scale_t

    # ------------------------------------------------------------------ locality pass
    perm = None        # set on a graph returned by reorder(): row i of this graph is row perm[i] of the caller's graph
    inv_perm = None    # and caller node v is row inv_perm[v] here

    def reorder(self, method="lpa", seed=0, sweeps=8):
        """(relabelled CSRGraph, perm): the engine's one-off locality pass (dgll_amd/reorder.py) -- community relabelling
        computed on the device the graph lives on.  Permute node data once with `x[perm]` (`to_engine_order`); results come
        back in the caller's node order through `to_caller_order`."""
        from . import reorder as _reorder

        if self.n_rows != self.n_cols:
            raise ValueError("reorder needs a square adjacency (one id space for rows and columns)")
        if method == "lpa" and self.nnz >= _reorder._LARGE_NNZ:
            method = "degree"         # label propagation sorts the edge list in one call; beyond 2^30 edges: hubs first (what
                                      # "lpa" degenerates to on a structure-free graph).  RMAT-25 / 27, F = 128 bf16, final
                                      # round-2 kernels: hubs-first 86 / 86 %, random 67 / 70 %, generator ids 41 / 53 % of 8 TB/s
        perm = _reorder.locality_order(self.rowptr, self.col, self.n_rows, method=method, seed=seed, sweeps=sweeps)
        return _reorder.relabel(self, perm), perm

    def to_engine_order(self, x):
        """Node-major tensor in the caller's order -> this (reordered) graph's row order."""
        return x if self.perm is None else x[self.perm.to(x.device)]

    def to_caller_order(self, y):
        """Rows of a result computed on this (reordered) graph -> the caller's node order."""
        return y if self.inv_perm is None else y[self.inv_perm.to(y.device)]

    # ------------------------------------------------------------------ HIP schedule
    def plan(self):
        """Opaque dgll_csr_plan* (long-row chunking for load balance); created on first use."""
        if not self.is_cuda:
            raise RuntimeError("CSRGraph.plan() needs a graph resident on the GPU")
        if self._plan is None:
            owner = getattr(self, "_structure_owner", None)
            if owner is not None:
                self._plan = owner.plan()
                return self._plan
            handle = C.c_void_p()
            with torch.cuda.device(self.device):
                stream = torch.cuda.current_stream(self.device).cuda_stream
                # max_degree (set by the block constructors) <= 128: no row can be "long" -> host-only plan, no sync.  A PADDED
                # block (graphs.PaddedBlock) rewrites its row pointers under a captured step: a chunk schedule derived from the
                # capture-time pointers would be stale on every replay, so it gets the host-only plan whatever its fan-out
                # (every row gathered inline; correct for any row length)
                threshold = -1 if ((self.max_degree is not None and self.max_degree <= 128) or getattr(self, "padded", False)) else 0
                _lib.check(_lib.lib.dgll_hip_csr_plan_create(stream, self.rowptr.data_ptr(), self.n_rows, self.nnz, threshold,
                                                             C.byref(handle)), "dgll_hip_csr_plan_create")
            self._plan = handle.value
            self._plan_finalizer = weakref.finalize(self, _lib.lib.dgll_hip_csr_plan_destroy, C.c_void_p(handle.value))
        return self._plan

    def workspace_bytes(self, feat):
        return int(_lib.lib.dgll_hip_csr_plan_workspace_bytes(self.plan(), int(feat)))

    def num_long_rows(self):
        return int(_lib.lib.dgll_hip_csr_plan_num_long_rows(self.plan()))


# ---------------------------------------------------------------------- adjacency cache
# The reference builds `adj` once and passes the same tensor to every layer call (utils.py:179 ->
# gcnconv.py:53-58).  Convert it once: key on the tensor object and validate with storage pointers/versions.
_adj_cache = {}


def as_csr_graph(adj):
    """CSRGraph for anything the reference's layers accept as an adjacency."""
    if isinstance(adj, CSRGraph):
        return adj
    if not isinstance(adj, torch.Tensor):
        raise TypeError("adjacency must be a CSRGraph or a torch tensor, got %r" % type(adj))
    key = id(adj)
    if adj.layout == torch.sparse_coo:
        stamp = (adj._indices().data_ptr(), adj._values().data_ptr(), adj._values()._version, tuple(adj.shape))
    elif adj.layout == torch.sparse_csr:
        stamp = (adj.crow_indices().data_ptr(), adj.values().data_ptr(), adj.values()._version, tuple(adj.shape))
    else:
        stamp = (adj.data_ptr(), adj._version, tuple(adj.shape))
    hit = _adj_cache.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    g = CSRGraph.from_dense(adj) if adj.layout == torch.strided else CSRGraph.from_torch_sparse(adj)
    _adj_cache[key] = (stamp, g)
    try:
        weakref.finalize(adj, _adj_cache.pop, key, None)
    except TypeError:
        pass
    return g
