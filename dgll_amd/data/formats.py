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


# ---------------------------------------------------------------------------------------------- raw edge-list datasets
# ogbn-products / Reddit as the reference's GPU-Accelerator scripts consume them come through DGL / OGB loaders
# (`dgll.data.RedditDataset`, `DglNodePropPredDataset`: /root/reference/dgll/GPU Accelerator/MQGCN.py:167-185,
# FeatureCache/*.py), neither of which is installable here.  What those loaders read from disk are plain arrays; the
# readers below take the same files directly, so bench.py / the examples run the NAMED dataset whenever it is present.

def _pick(d, names, what, required=True):
    for k in names:
        if k in d:
            return d[k]
    if required:
        raise KeyError("%s: none of %s found (have: %s)" % (what, ", ".join(names), ", ".join(sorted(d.keys()))))
    return None


def _csr_from_edges(src, dst, n, symmetrise):
    """Sorted, de-duplicated CSR arrays (int64 indptr, int64 indices) of the directed edge list src -> dst, row = dst's
    in-neighbour list is NOT what the samplers want: row v lists v's neighbours `u` for edges (v, u), as DGraph.edges."""
    src = np.asarray(src, dtype=np.int64).ravel()
    dst = np.asarray(dst, dtype=np.int64).ravel()
    if src.shape != dst.shape:
        raise ValueError("edge endpoints differ in length")
    if src.size and (min(src.min(), dst.min()) < 0 or max(src.max(), dst.max()) >= n):
        raise ValueError("edge endpoint outside [0, %d)" % n)
    if symmetrise:
        src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
    key = np.unique(src * n + dst)
    row = key // n
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(row, minlength=n), out=indptr[1:])
    return indptr, key - row * n


def load_edge_list_npz(path, symmetrise=False):
    """One .npz holding a whole node-classification dataset (the dict OGB's NodePropPredDataset builds from
    raw/edge.csv.gz + node-feat.csv.gz + node-label.csv.gz, saved with np.savez; also PyG / hand-made dumps):
       edges:    `edge_index` [2, E]  |  `src` + `dst`  |  `row` + `col`  |  scipy CSR `indptr` + `indices` (+ `shape`)
       features: `node_feat` | `feat` | `features` | `x`          labels: `node_label` | `label` | `labels` | `y`
       splits (optional): `train_idx` / `valid_idx` / `test_idx`  or boolean `train_mask` / `val_mask` / `test_mask`
    -> DGraph with CSR-backed adjacency lists.  ogbn-products' edge list is undirected-once: pass symmetrise=True."""
    from .dgraph import DGraph

    with np.load(path, allow_pickle=False) as z:
        d = {k: z[k] for k in z.files}
    feat = _pick(d, ("node_feat", "feat", "features", "x"), "node features")
    label = _pick(d, ("node_label", "label", "labels", "y"), "node labels")
    n = int(feat.shape[0])
    if "indptr" in d and "indices" in d:
        indptr, indices = d["indptr"].astype(np.int64), d["indices"].astype(np.int64)
        if indptr.shape[0] != n + 1:
            raise ValueError("CSR indptr has %d entries for %d nodes" % (indptr.shape[0], n))
        if symmetrise:
            row = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
            indptr, indices = _csr_from_edges(row, indices, n, True)
    else:
        if "edge_index" in d:
            ei = d["edge_index"]
            ei = ei if ei.shape[0] == 2 else ei.T
            src, dst = ei[0], ei[1]
        else:
            src = _pick(d, ("src", "row"), "edge sources")
            dst = _pick(d, ("dst", "col"), "edge destinations")
        indptr, indices = _csr_from_edges(src, dst, n, symmetrise)
    label = np.asarray(label).reshape(n, -1)
    label = label[:, 0] if label.shape[1] == 1 else label
    masks = {}
    for name, idx_keys, mask_keys in (("train_mask", ("train_idx", "train"), ("train_mask",)),
                                      ("validation_mask", ("valid_idx", "val_idx", "valid"), ("val_mask", "valid_mask", "validation_mask")),
                                      ("test_mask", ("test_idx", "test"), ("test_mask",))):
        m = _pick(d, mask_keys, name, required=False)
        if m is None:
            idx = _pick(d, idx_keys, name, required=False)
            if idx is not None:
                m = np.zeros(n, dtype=bool)
                m[np.asarray(idx, dtype=np.int64)] = True
        if m is not None:
            masks[name] = torch.from_numpy(np.asarray(m, dtype=bool))
    lab_t = torch.from_numpy(label.astype(np.int64) if np.issubdtype(label.dtype, np.integer) or label.ndim == 1 else label.astype(np.float32))
    return DGraph.from_csr(indptr, indices, labels=lab_t, features=torch.from_numpy(np.ascontiguousarray(feat, dtype=np.float32)), **masks)


def load_reddit(data_dir):
    """Reddit as DGL's RedditDataset stores it (the dataset the reference's MQ-GNN scripts train on, README.md:40-49):
    `reddit_data.npz` (`feature` [232965, 602], `label`, `node_types` 1 / 2 / 3 = train / val / test) and
    `reddit_graph.npz` (scipy sparse COO: `row`, `col`, `data`, `shape`; or CSR: `indptr`, `indices`).  -> DGraph."""
    from .dgraph import DGraph

    with np.load(os.path.join(data_dir, "reddit_data.npz"), allow_pickle=False) as z:
        feat, label, types = z["feature"], z["label"], z["node_types"]
    n = int(feat.shape[0])
    with np.load(os.path.join(data_dir, "reddit_graph.npz"), allow_pickle=False) as z:
        if "indptr" in z.files:
            indptr, indices = z["indptr"].astype(np.int64), z["indices"].astype(np.int64)
        else:
            indptr, indices = _csr_from_edges(z["row"], z["col"], n, False)
    return DGraph.from_csr(indptr, indices, labels=torch.from_numpy(label.astype(np.int64)),
                           features=torch.from_numpy(np.ascontiguousarray(feat, dtype=np.float32)),
                           train_mask=torch.from_numpy(types == 1), validation_mask=torch.from_numpy(types == 2),
                           test_mask=torch.from_numpy(types == 3))


def load_ogb_raw(raw_dir, symmetrise=True):
    """An OGB node-property dataset from its RAW csv files (what `ogb` downloads before processing): `edge.csv[.gz]`,
    `node-feat.csv[.gz]`, `node-label.csv[.gz]` in `raw_dir`, and, when present next to it, `../split/*/{train,valid,test}
    .csv[.gz]`.  ogbn-products: 2 449 029 nodes, 61 859 140 undirected edges, 100 features, 47 classes.  -> DGraph."""
    import gzip

    from .dgraph import DGraph

    def table(stem, dtype, required=True):
        for ext in (".csv.gz", ".csv"):
            p = os.path.join(raw_dir, stem + ext)
            if os.path.exists(p):
                opener = gzip.open if ext.endswith(".gz") else open
                with opener(p, "rt") as f:
                    return np.loadtxt(f, delimiter=",", dtype=dtype, ndmin=2)
        if required:
            raise FileNotFoundError("%s.csv[.gz] not found under %s" % (stem, raw_dir))
        return None

    feat = table("node-feat", np.float32)
    label = table("node-label", np.int64)
    edge = table("edge", np.int64)
    n = int(feat.shape[0])
    indptr, indices = _csr_from_edges(edge[:, 0], edge[:, 1], n, symmetrise)
    masks = {}
    split_root = os.path.join(os.path.dirname(os.path.abspath(raw_dir)), "split")
    if os.path.isdir(split_root):
        for sub in sorted(os.listdir(split_root)):
            for name, stem in (("train_mask", "train"), ("validation_mask", "valid"), ("test_mask", "test")):
                for ext in (".csv.gz", ".csv"):
                    p = os.path.join(split_root, sub, stem + ext)
                    if os.path.exists(p) and name not in masks:
                        opener = gzip.open if ext.endswith(".gz") else open
                        with opener(p, "rt") as f:
                            idx = np.loadtxt(f, delimiter=",", dtype=np.int64).ravel()
                        m = np.zeros(n, dtype=bool)
                        m[idx] = True
                        masks[name] = torch.from_numpy(m)
            break
    return DGraph.from_csr(indptr, indices, labels=torch.from_numpy(label[:, 0]), features=torch.from_numpy(feat), **masks)


def load_node_dataset(path, **kw):
    """Dispatch on what `path` is: a .npz file (load_edge_list_npz), a directory with reddit_data.npz (load_reddit) or an
    OGB raw directory (load_ogb_raw)."""
    if os.path.isfile(path):
        return load_edge_list_npz(path, **kw)
    if os.path.exists(os.path.join(path, "reddit_data.npz")):
        return load_reddit(path)
    if os.path.isdir(os.path.join(path, "raw")):
        path = os.path.join(path, "raw")
    return load_ogb_raw(path, **kw)
