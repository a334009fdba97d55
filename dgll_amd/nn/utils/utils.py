"""Host-side helpers of the reference's dgll/nn/utils/utils.py (and the duplicate dgll/nn/utililities.py:14-47) that
sit directly in front of the GCN/GAT layers: the Cora-format loader and the adjacency hand-over.

    encode_onehot                      utils.py:70-76
    load_data                          utils.py:146-185
    normalize                          utils.py:240-247
    sparse_mx_to_torch_sparse_tensor   utils.py:250-257
    accuracy                           utils.py:260-264

The sampling/k-hop helpers of that file (createIndex, load_khop, ...) belong to a different workflow and are not
mirrored.  `load_data` builds the adjacency with dgll_amd.prep (torch ops, no scipy round trip) and returns the same
tuple as the reference; `adj` is the reference's hand-over type (fp32 sparse COO, indices sorted) which every layer
here accepts and converts to CSR once."""
import numpy as np
import torch

from ... import prep
from ...data.formats import load_citation


def encode_onehot(labels):
    """One-hot rows for an array of class names.  The reference enumerates `set(labels)` (utils.py:71), whose order
    changes with the interpreter's string-hash seed; here classes are numbered in sorted order, so runs agree."""
    classes = sorted(set(labels.tolist() if hasattr(labels, "tolist") else labels))
    index = {c: i for i, c in enumerate(classes)}
    out = np.zeros((len(labels), len(classes)), dtype=np.int32)
    out[np.arange(len(labels)), [index[c] for c in (labels.tolist() if hasattr(labels, "tolist") else labels)]] = 1
    return out


def normalize(mx):
    """Row-normalise a scipy sparse matrix or a 2-D array; all-zero rows stay zero (utils.py:240-247)."""
    import scipy.sparse as sp

    rowsum = np.asarray(mx.sum(1)).flatten().astype(np.float64)
    with np.errstate(divide="ignore"):
        r_inv = np.power(rowsum, -1.0)
    r_inv[np.isinf(r_inv)] = 0.0
    if sp.issparse(mx):
        return sp.diags(r_inv).dot(mx)
    return np.asarray(mx) * r_inv[:, None]


def sparse_mx_to_torch_sparse_tensor(sparse_mx):
    """scipy sparse -> fp32 torch sparse COO (utils.py:250-257; the deprecated F.sparse.FloatTensor ctor is replaced by
    sparse_coo_tensor, same contents)."""
    coo = sparse_mx.tocoo().astype(np.float32)
    indices = torch.from_numpy(np.vstack((coo.row, coo.col)).astype(np.int64))
    return torch.sparse_coo_tensor(indices, torch.from_numpy(coo.data), torch.Size(coo.shape))


def accuracy(output, labels):
    """utils.py:260-264."""
    preds = output.max(1)[1].type_as(labels)
    return preds.eq(labels).double().sum() / len(labels)


def load_data(path="./Datasets/Cora/", dataset="cora", device="cpu"):
    """(adj, features, labels, idx_train, idx_val, idx_test) of a Cora-format dataset -- utils.py:146-185:
    adj = D^-1 (max(A, A^T) + I) with repeated citations summed first (scipy's coo -> csr does that, :160-164,171),
    features row-normalised (:170), labels = class index, the fixed 140 / 300 / 1000 split (:173-175)."""
    feats, names, row, col = load_citation(path, dataset)
    n = feats.shape[0]
    key, counts = np.unique(row * n + col, return_counts=True)
    r = torch.from_numpy(key // n).to(device)
    c = torch.from_numpy(key % n).to(device)
    graph = prep.normalized_adjacency(r, c, n, val=torch.from_numpy(counts.astype(np.float32)).to(device))
    adj = prep.to_torch_coo(graph)
    features = torch.from_numpy(normalize(feats).astype(np.float32)).to(device)
    labels = torch.from_numpy(np.where(encode_onehot(names))[1]).long().to(device)
    idx_train = torch.arange(140, device=device)
    idx_val = torch.arange(200, 500, device=device)
    idx_test = torch.arange(500, 1500, device=device)
    return adj, features, labels, idx_train, idx_val, idx_test
