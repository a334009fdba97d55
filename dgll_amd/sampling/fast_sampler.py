"""FastNeighborSampler -- the reference's multi-hop sampler (dgllsampler.py:10-21) with the per-seed loop in native
code (dgll_host_sample_neighbors), bit-exact with DGLLNeighborSampler / the reference under `random.seed(s)`.

The Python side keeps a CSR copy of DGraph.edges, hands the interpreter's Mersenne-Twister state to the native call and
stores the advanced state back, so it can be freely interleaved with other users of the global `random` module.
"""
import ctypes as C
import math
import random

import numpy as np
import torch

from .. import _lib
from .base_sampler import Base_sampler, sugbraph


def _setsize(k):
    """CPython 3.10 random.sample: threshold between the pool and the set algorithm."""
    setsize = 21
    if k > 5:
        setsize += 4 ** math.ceil(math.log(k * 3, 4))
    return setsize


class _LazyInput:
    """Stands for `subgs[0].src_nodes()` (the input nodes of the batch) without forcing the deferred translation."""

    def __init__(self, subg):
        self._subg = subg

    def resolve(self):
        return self._subg.src_nodes()


class FastNeighborSampler(Base_sampler):
    def __init__(self, fanouts, defer_last_hop=False):
        super().__init__()
        self.fanouts = fanouts
        self.defer_last_hop = defer_last_hop
        self._csr_of = None
        self._indptr = self._indices = None

    def _csr(self, g):
        if hasattr(g.edges, "indptr"):           # CSRAdjacency: use the arrays directly
            return (np.ascontiguousarray(g.edges.indptr, dtype=np.int64), np.ascontiguousarray(g.edges.indices, dtype=np.int64))
        if self._csr_of is not g.edges:
            deg = np.fromiter((len(e) for e in g.edges), dtype=np.int64, count=len(g.edges))
            self._indptr = np.zeros(len(g.edges) + 1, dtype=np.int64)
            np.cumsum(deg, out=self._indptr[1:])
            self._indices = np.fromiter((u for e in g.edges for u in e), dtype=np.int64, count=int(self._indptr[-1]))
            self._csr_of = g.edges
        return self._indptr, self._indices

    def sample_neighbours(self, g, nodes, fanout=None, defer_translation=False):
        indptr, indices = self._csr(g)
        seeds = np.ascontiguousarray(nodes.numpy() if isinstance(nodes, torch.Tensor) else np.asarray(nodes), dtype=np.int64)
        deg = indptr[seeds + 1] - indptr[seeds]
        cap = int(deg.sum() if fanout is None else np.minimum(deg, fanout).sum())
        src = np.empty(cap, dtype=np.int64)
        dst = np.empty(cap, dtype=np.int64)
        counts = np.empty(len(seeds), dtype=np.int64)
        version, internal, gauss = random.getstate()
        state = np.array(internal[:624], dtype=np.uint32)
        index = C.c_int(internal[624])
        n_out = C.c_int64(0)
        code = _lib.lib.dgll_host_sample_neighbors(
            state.ctypes.data, C.byref(index), indptr.ctypes.data, indices.ctypes.data, seeds.ctypes.data, len(seeds),
            -1 if fanout is None else int(fanout), _setsize(fanout) if fanout is not None else 21, src.ctypes.data,
            None if defer_translation else dst.ctypes.data, counts.ctypes.data, cap, C.byref(n_out))
        _lib.check(code, "dgll_host_sample_neighbors")
        random.setstate((version, tuple(int(x) for x in state) + (index.value,), gauss))
        ptr = np.zeros(len(seeds) + 1, dtype=np.int64)
        np.cumsum(counts, out=ptr[1:])
        finish = None
        if defer_translation:     # positions -> ids later (no generator state involved): whoever first reads the ids pays

            def finish():
                _lib.check(_lib.lib.dgll_host_translate_neighbors(indptr.ctypes.data, indices.ctypes.data, seeds.ctypes.data,
                                                                  len(seeds), counts.ctypes.data, src.ctypes.data,
                                                                  dst.ctypes.data), "dgll_host_translate_neighbors")

        sg = sugbraph(torch.from_numpy(src), torch.from_numpy(dst), torch.from_numpy(ptr), finish=finish)
        sg.max_degree = None if fanout is None else int(fanout)      # lets to_block() skip the long-row scan (host-only plan)
        return sg

    def sample(self, g, seed_nodes):
        output_nodes = seed_nodes
        subgs = []
        order = list(reversed(self.fanouts))
        for k, fanout in enumerate(order):
            last = k == len(order) - 1
            # the outermost hop's ids feed nothing inside this call: leave their translation to the first reader (the
            # pipeline's loading thread), so this thread -- the only one allowed to touch the generator -- moves on
            subg = self.sample_neighbours(g, seed_nodes, fanout, defer_translation=last and self.defer_last_hop)
            subgs.insert(0, subg)
            if not last:
                seed_nodes = subg.src_nodes()
        return _LazyInput(subgs[0]) if self.defer_last_hop else subgs[0].src_nodes(), output_nodes, subgs
