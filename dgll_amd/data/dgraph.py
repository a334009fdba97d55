"""DGraph -- the reference's in-memory graph container (/root/reference/dgll/data/dgraph.py:18-132).

Same constructor and query methods (`get_neighbors`, `get_induced_subgraph`, `get_labels`, `get_features`,
`get_train_nodes`, `get_validation_nodes`, `get_test_nodes`): `edges` is a python list of adjacency lists, features /
labels / masks are tensors.  Additions the reference's own scripts expect but the class lacks: `feature_size()`
(graphage.py:28) and `to_csr()` (hand-over to the aggregation engine).
"""
import torch

from ..graph import CSRGraph


class CSRAdjacency:
    """List-of-lists view over CSR arrays (numpy int64 indptr / indices): `edges[v]` is node v's neighbour list, as
    DGraph.edges is in the reference (dgraph.py:33), without materialising a hundred million Python ints."""

    def __init__(self, indptr, indices):
        self.indptr, self.indices = indptr, indices

    def __len__(self):
        return len(self.indptr) - 1

    def __getitem__(self, v):
        return self.indices[self.indptr[v]:self.indptr[v + 1]].tolist()

    def __iter__(self):
        for v in range(len(self)):
            yield self[v]


class DGraph(object):
    def __init__(self, nodes=None, edges=None, labels=None, features=None, train_mask=None, test_mask=None,
                 validation_mask=None):
        self.nodes = nodes
        self.edges = edges
        self.labels = labels
        self.features = features
        self.train_mask = train_mask
        self.test_mask = test_mask
        self.validation_mask = validation_mask

    # ---- queries (dgraph.py:49-132) ------------------------------------------------------------------
    def get_neighbors(self, nodes):
        """Adjacency lists of `nodes` (a tensor of ids), in order -- dgraph.py:59-62."""
        ids = nodes.tolist() if hasattr(nodes, "tolist") else list(nodes)
        return [self.edges[i] for i in ids]

    def get_induced_subgraph(self, nodes):
        """Dense int32 adjacency of the subgraph induced by `nodes` -- dgraph.py:74-81."""
        ids = nodes.tolist()
        position = {v: i for i, v in enumerate(ids)}
        result = torch.zeros(len(ids), len(ids), dtype=torch.int32)
        for v in ids:
            cols = [position[u] for u in self.edges[v] if u in position]
            if cols:
                result[position[v], cols] = 1
        return result

    def get_labels(self, nodes):
        return self.labels[nodes]

    def get_features(self, nodes):
        return self.features[nodes]

    def get_train_nodes(self):
        return self.nodes[self.train_mask]

    def get_validation_nodes(self):
        return self.nodes[self.validation_mask]

    def get_test_nodes(self):
        return self.nodes[self.test_mask]

    # ---- additions -----------------------------------------------------------------------------------
    @classmethod
    def from_csr(cls, indptr, indices, **kw):
        """DGraph whose adjacency lists are backed by CSR arrays (numpy int64)."""
        n = len(indptr) - 1
        kw.setdefault("nodes", torch.arange(n))
        return cls(edges=CSRAdjacency(indptr, indices), **kw)

    def feature_size(self):
        return int(self.features.shape[1])

    def num_nodes(self):
        return len(self.edges)

    def to_csr(self, device="cpu", weighted=False):
        """Whole adjacency as a CSRGraph (row v gathers from edges[v]), optionally with D^-1 weights."""
        if isinstance(self.edges, CSRAdjacency):
            rowptr = torch.from_numpy(self.edges.indptr.astype("int64"))
            col = torch.from_numpy(self.edges.indices.astype("int32"))
            deg = rowptr[1:] - rowptr[:-1]
        else:
            deg = torch.tensor([len(e) for e in self.edges], dtype=torch.int64)
            rowptr = torch.zeros(len(self.edges) + 1, dtype=torch.int64)
            torch.cumsum(deg, 0, out=rowptr[1:])
            col = torch.tensor([u for e in self.edges for u in e], dtype=torch.int32)
        val = (1.0 / deg.clamp(min=1).float())[torch.repeat_interleave(torch.arange(len(self.edges)), deg)] if weighted else None
        return CSRGraph(rowptr, col, val, len(self.edges), len(self.edges)).to(device)
