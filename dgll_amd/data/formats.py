"""On-disk graph formats the reference reads before its layers run, as vectorised numpy:

* GraphSAGE-format PPI -- `{split}_graph.json` (networkx node-link), `{split}_feats.npy`, `{split}_labels.npy`,
  `{split}_graph_id.npy` -- Evaluation/PPI/ppi_dataloader.py:10-61;
* Cora-format citation text -- `<name>.content` (id, features..., class string) and `<name>.cites` (cited, citing id
  pairs) -- dgll/nn/utils/utils.py:146-185.

Both loaders return host tensors; `.to("cuda")`/CSRGraph construction is the caller's next line, as in the reference's
training scripts (Evaluation/PPI/train_gcn.py:36-39).
"""
import json
import os

import numpy as np
import torch


def _node_link_edges(path):
    """Directed, de-duplicated (src, dst) pairs of a node-link JSON file, as nx.DiGraph(node_link_graph(.)) holds them
    (ppi_dataloader.py:22-23): an undirected file contributes both directions, parallel links collapse."""
    with open(path, "r") as f:
        data = json.load(f)
    links = data.get("links", data.get("edges", []))
    ids = [n["id"] for n in data["nodes"]]
    if ids != list(range(len(ids))):
        raise ValueError("node ids must be 0..N-1 in file order (they index the rows of *_feats.npy)")
    e = np.array([(l["source"], l["target"]) for l in links], dtype=np.int64).reshape(-1, 2)
    if not data.get("directed", False):
        e = np.concatenate([e, e[:, ::-1]], axis=0)
    n = len(ids)
    if e.size and (e.min() < 0 or e.max() >= n):
        raise ValueError("link endpoint outside the node list")
    key = np.unique(e[:, 0] * n + e[:, 1])          # sorted by (src, dst): a canonical edge order
    return key // n, key % n, n


def load_ppi_dataset(data_dir, split):
    """[(edge_index int64 [2, E], x fp32 [n, F], y fp32 [n, C]) per graph of the split] -- ppi_dataloader.py:10-61.

    Differences from the reference, none of which changes a model output: edges come sorted by (src, dst) instead of in
    networkx iteration order (the adjacency is assembled by summation, gcn_model.py:56); node ids are shifted by the
    graph's first node rather than by the smallest edge endpoint (:53 -- the two differ only when a graph's first node
    has no edge, where the reference mis-aligns edges and feature rows)."""
    src, dst, n = _node_link_edges(os.path.join(data_dir, "%s_graph.json" % split))
    x = torch.from_numpy(np.load(os.path.join(data_dir, "%s_feats.npy" % split))).float()
    y = torch.from_numpy(np.load(os.path.join(data_dir, "%s_labels.npy" % split))).float()
    gid = np.load(os.path.join(data_dir, "%s_graph_id.npy" % split)).astype(np.int64)
    if not (x.shape[0] == y.shape[0] == gid.shape[0] == n):
        raise ValueError("feats/labels/graph_id/graph.json disagree on the node count")
    gid = gid - gid.min()                                                     # :39
    graphs = []
    for i in range(int(gid.max()) + 1 if n else 0):
        nodes = np.nonzero(gid == i)[0]
        if nodes.size == 0:
            raise ValueError("graph id %d has no nodes" % i)
        if nodes[-1] - nodes[0] + 1 != nodes.size:
            raise ValueError("the nodes of graph %d are not contiguous rows" % i)
        keep = (gid[src] == i) & (gid[dst] == i) & (src != dst)               # induced subgraph :48, no self-loops :56
        ei = torch.from_numpy(np.stack([src[keep] - nodes[0], dst[keep] - nodes[0]]))
        sel = torch.from_numpy(nodes)
        graphs.append((ei, x[sel], y[sel]))
    return graphs


def load_citation(path, dataset):
    """Parse `<path><dataset>.content` / `.cites` (utils.py:151-163): returns (features fp32 [n, F] (raw), class strings
    [n], directed edges (row, col) int64 in file order re-indexed to content-file row order)."""
    table = np.genfromtxt("%s%s.content" % (path, dataset), dtype=np.dtype(str))
    if table.ndim != 2 or table.shape[1] < 3:
        raise ValueError("%s.content: expected rows of <id> <features...> <label>" % dataset)
    feats = table[:, 1:-1].astype(np.float32)
    ids = table[:, 0].astype(np.int64)
    order = np.argsort(ids, kind="stable")
    cites = np.genfromtxt("%s%s.cites" % (path, dataset), dtype=np.int64).reshape(-1, 2)
    pos = np.searchsorted(ids[order], cites.ravel())
    pos = np.clip(pos, 0, ids.size - 1)
    if not np.array_equal(ids[order][pos], cites.ravel()):
        raise ValueError("%s.cites names a paper id that %s.content does not list" % (dataset, dataset))
    edges = order[pos].reshape(-1, 2)
    return feats, table[:, -1], edges[:, 0], edges[:, 1]
