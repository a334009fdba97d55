"""Neighbour sampling base classes (/root/reference/dgll/sampling/base_sampler.py:4-109).

The drawn IDs must be bit-identical to the reference's under `random.seed(s)` (BASELINE north_star).  That is obtained
by construction: the same stdlib call -- `random.sample(neighbors, fanout)`, only when `len(neighbors) > fanout` -- is
made for the same seeds in the same order (base_sampler.py:45-58), so the Mersenne-Twister stream is consumed
identically.  `sugbraph` keeps the reference's COO view (`src_data`, `dst_data`, `nodes()`) and additionally remembers
how many neighbours each seed OCCURRENCE received (`indptr`), which is what the CSR block needs (two occurrences of
the same seed are indistinguishable in the COO pair).
"""
import random

import torch

from ..graph import CSRGraph


class sugbraph():
    """Sampled one-hop subgraph: edge k goes src_data[k] -> dst_data[k] (base_sampler.py:65-109)."""

    def __init__(self, src_data, dst_data, indptr=None, finish=None):
        """finish: optional callable completing src_data / dst_data in place on first access (the native sampler can hand
        the outermost hop over with neighbour POSITIONS only and translate them to ids later, on the loading thread)."""
        self._src, self._dst, self._finish = src_data, dst_data, finish
        self.indptr = indptr
        self._graph_nodes = None     # unique(src ++ dst), base_sampler.py:82 -- computed on first use (it costs more than
                                     # drawing the sample itself on multi-million-edge hops)

    def _complete(self):
        if self._finish is not None:
            finish, self._finish = self._finish, None
            finish()

    @property
    def src_data(self):
        self._complete()
        return self._src

    @property
    def dst_data(self):
        self._complete()
        return self._dst

    @property
    def graph_nodes(self):
        if self._graph_nodes is None:
            self._graph_nodes = torch.unique(torch.cat((self.src_data, self.dst_data)))
        return self._graph_nodes

    def src_nodes(self):
        return self.src_data

    def dst_nodes(self):
        return self.dst_data

    def nodes(self):
        return self.graph_nodes

    def num_src_nodes(self):
        return self._src.shape[0]

    def num_dst_nodes(self):
        return self._src.shape[0] if self._dst is None else self._dst.shape[0]

    def get_features(self, g, subgs):
        unique_nodes = torch.unique(torch.cat([subg.nodes() for subg in subgs]))
        return g.get_features(unique_nodes)

    def to_block(self, device):
        """CSR block for the aggregation kernels: row r = r-th seed occurrence, its columns index this layer's
        `src_data` positions (the next hop's feature rows are stored in that order, so col = arange)."""
        if self.indptr is None:
            raise ValueError("this sugbraph was not produced by a sampler (no per-seed counts)")
        n_src = int(self._src.shape[0])
        g = CSRGraph(self.indptr.to(device), torch.arange(n_src, dtype=torch.int32, device=device), None,
                     int(self.indptr.numel() - 1), n_src, check=False)
        g.identity_cols = True           # the transposed product of the backward pass is then a plain row gather
        # row-length bound: the sampler's fan-out when it recorded one, else a single-threaded numpy pass over the host
        # indptr (a torch CPU reduction here would wake the intra-op thread pool next to the sampler thread: measured 2.4x
        # slower sampling)
        bound = getattr(self, "max_degree", None)
        if bound is None and self.indptr.numel() > 1 and not self.indptr.is_cuda:
            import numpy as np

            bound = int(np.diff(self.indptr.numpy()).max())
        g.max_degree = bound
        return g


class Base_sampler(object):
    """Subclass and override `sample(g, nodes)` (base_sampler.py:4-28)."""

    def sample(self, g, nodes):
        raise NotImplementedError

    def _subgraph(self, nodes, neighbors_list):
        seeds = nodes.tolist()
        src_list, dst_list, counts = [], [], []
        for seed, neighbors in zip(seeds, neighbors_list):
            src_list.extend(neighbors)
            dst_list.extend([seed] * len(neighbors))
            counts.append(len(neighbors))
        indptr = torch.zeros(len(seeds) + 1, dtype=torch.int64)
        torch.cumsum(torch.tensor(counts, dtype=torch.int64), 0, out=indptr[1:])
        return sugbraph(torch.tensor(src_list, dtype=torch.int64), torch.tensor(dst_list, dtype=torch.int64), indptr)

    def sample_neighbours(self, g, nodes, fanout=None):
        random_neighbors = []
        for neighbors in g.get_neighbors(nodes):
            if len(neighbors) == 0:
                random_neighbors.append([])
            elif fanout is None or len(neighbors) <= fanout:
                random_neighbors.append(neighbors)                       # everything, no RNG draw
            else:
                random_neighbors.append(random.sample(neighbors, fanout))  # base_sampler.py:56
        return self._subgraph(nodes, random_neighbors)

    def get_adj(self, g, subgs):
        unique_nodes = torch.unique(torch.cat([subg.nodes() for subg in subgs]))
        return g.get_induced_subgraph(unique_nodes)
