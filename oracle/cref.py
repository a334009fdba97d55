"""numpy front-end of oracle.c (CPU restatement; test infrastructure only).

Every wrapper names the reference lines it restates; see oracle.c for the arithmetic.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    """Compile oracle.c with gcc (see oracle/Makefile)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _csr(rowptr, col, val=None):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val = None if val is None else _f32(val)
    return rowptr, col, val


def num_threads():
    return lib().oracle_num_threads()


def set_num_threads(n):
    lib().oracle_set_num_threads(C.c_int(int(n)))


def spmm_csr(rowptr, col, val, X, reduce="sum"):
    """Y = A.X  -- gcnconv.py:31 (F.spmm), PPI/gcn_model.py:76; reduce='mean' -- sageconv.py:33-34."""
    rowptr, col, val = _csr(rowptr, col, val)
    X = _f32(X)
    n_rows, feat = rowptr.shape[0] - 1, X.shape[1]
    Y = np.empty((n_rows, feat), dtype=np.float32)
    lib().oracle_spmm_csr_f32(_p(rowptr), _p(col), _p(val), _p(X), C.c_int64(X.shape[1]), _p(Y),
                              C.c_int64(feat), C.c_int64(n_rows), C.c_int(feat),
                              C.c_int({"sum": 0, "mean": 1}[reduce]))
    return Y


def spmm_coo(row, col, val, X, n_rows):
    """Uncoalesced COO product -- PPI/gcn_model.py:56,76."""
    row = np.ascontiguousarray(row, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int64)
    val = None if val is None else _f32(val)
    X = _f32(X)
    feat = X.shape[1]
    Y = np.empty((n_rows, feat), dtype=np.float32)
    lib().oracle_spmm_coo_f32(_p(row), _p(col), _p(val), C.c_int64(row.shape[0]), _p(X), C.c_int64(feat),
                              _p(Y), C.c_int64(feat), C.c_int64(n_rows), C.c_int(feat))
    return Y


def spmm_csr_max(rowptr, col, X):
    """max over neighbours -- sageconv.py:37-38."""
    rowptr, col, _ = _csr(rowptr, col)
    X = _f32(X)
    n_rows, feat = rowptr.shape[0] - 1, X.shape[1]
    Y = np.empty((n_rows, feat), dtype=np.float32)
    lib().oracle_spmm_csr_max_f32(_p(rowptr), _p(col), _p(X), C.c_int64(feat), _p(Y), C.c_int64(feat),
                                  C.c_int64(n_rows), C.c_int(feat))
    return Y


def sddmm_csr(rowptr, col, G, B):
    """grad_values -- gatconv.py:76-78."""
    rowptr, col, _ = _csr(rowptr, col)
    G, B = _f32(G), _f32(B)
    out = np.empty(col.shape[0], dtype=np.float32)
    lib().oracle_sddmm_csr_f32(_p(rowptr), _p(col), _p(G), C.c_int64(G.shape[1]), _p(B), C.c_int64(B.shape[1]),
                               _p(out), C.c_int64(rowptr.shape[0] - 1), C.c_int(G.shape[1]))
    return out


def gemm(A, B, bias=None):
    """C = A.B (+bias) -- gcnconv.py:30, sageconv.py:41,72, gatconv.py:31,117."""
    A, B = _f32(A), _f32(B)
    bias = None if bias is None else _f32(bias)
    M, K = A.shape
    N = B.shape[1]
    Cm = np.empty((M, N), dtype=np.float32)
    lib().oracle_gemm_f32(_p(A), C.c_int64(K), _p(B), C.c_int64(N), _p(Cm), C.c_int64(N),
                          C.c_int64(M), C.c_int(N), C.c_int(K), _p(bias))
    return Cm


def gat_fwd(rowptr, col, H, S, T, heads, alpha, apply_elu=True, mode=0, want_edges=False):
    """sparseGatConv (mode 0, gatconv.py:111-148) / gatConv on the nonzeros (mode 1, gatconv.py:30-54)."""
    rowptr, col, _ = _csr(rowptr, col)
    H, S, T = _f32(H), _f32(S), _f32(T)
    n, width = H.shape
    fo = width // heads
    out = np.empty((n, width), dtype=np.float32)
    edge_e = np.empty((col.shape[0], heads), dtype=np.float32) if want_edges else None
    rowsum = np.empty((n, heads), dtype=np.float32) if want_edges else None
    lib().oracle_gat_fwd_f32(_p(rowptr), _p(col), _p(H), C.c_int64(width), _p(S), _p(T), _p(out),
                             C.c_int64(width), _p(edge_e), _p(rowsum), C.c_int64(n), C.c_int(heads),
                             C.c_int(fo), C.c_float(alpha), C.c_int(int(apply_elu)), C.c_int(mode))
    return (out, edge_e, rowsum) if want_edges else out


def sage_fwd(src, nbr, Ws, Wn, aggr="mean", hid="sum", act=True):
    """sageConv.forward with the documented fix -- sageconv.py:32-45,70-83."""
    src, nbr, Ws, Wn = _f32(src), _f32(nbr), _f32(Ws), _f32(Wn)
    N, K, D = nbr.shape
    Hd = Ws.shape[1]
    out = np.empty((N, Hd * (2 if hid == "concat" else 1)), dtype=np.float32)
    lib().oracle_sage_fwd_f32(_p(src), _p(nbr), _p(Ws), _p(Wn), _p(out), C.c_int64(N), C.c_int(K), C.c_int(D),
                              C.c_int(Hd), C.c_int({"mean": 0, "sum": 1, "max": 2}[aggr]),
                              C.c_int(int(hid == "concat")), C.c_int(int(act)))
    return out


def gcn_fused_fwd(row_ptr, col_idx, values, X, W, actual_F):
    """relu(A.(X[:, :actual_F].W)) -- FusedKernel/gcn_fused_kernel.cu:39-69 (int32 CSR, :190-195)."""
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
    col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
    values, X, W = _f32(values), _f32(X), _f32(W)
    N, F_padded = X.shape
    H_dim = W.shape[1]
    out = np.empty((N, H_dim), dtype=np.float32)
    lib().oracle_gcn_fused_fwd_f32(_p(row_ptr), _p(col_idx), _p(values), _p(X), _p(W), _p(out), C.c_int(N),
                                   C.c_int(F_padded), C.c_int(actual_F), C.c_int(H_dim))
    return out


def gather_rows(X, idx):
    """features[nodes] -- data/dgraph.py:105."""
    X = _f32(X)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    out = np.empty((idx.shape[0], X.shape[1]), dtype=np.float32)
    lib().oracle_gather_rows_f32(_p(X), C.c_int64(X.shape[1]), _p(idx), C.c_int64(idx.shape[0]), _p(out),
                                 C.c_int64(X.shape[1]), C.c_int(X.shape[1]))
    return out


# ------------------------------------------------------------------------------------------------
# numpy-only helpers (format conversion used by the tests; restates nn/utils/utils.py:240-257)
# ------------------------------------------------------------------------------------------------
def coo_to_csr(row, col, val, n_rows):
    """Sort COO triples row-major (stable) and sum duplicates, as torch's coalesce() would."""
    row = np.asarray(row, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    val = np.ones(row.shape[0], np.float32) if val is None else np.asarray(val, dtype=np.float32)
    n_cols = int(col.max()) + 1 if col.size else 1
    key = row * max(n_cols, 1) + col
    order = np.argsort(key, kind="stable")
    key, val = key[order], val[order]
    uniq, start = np.unique(key, return_index=True)
    v = np.add.reduceat(val, start).astype(np.float32) if key.size else val
    r = uniq // max(n_cols, 1)
    c = (uniq % max(n_cols, 1)).astype(np.int32)
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.add.at(rowptr, r + 1, 1)
    return np.cumsum(rowptr), c, v


def row_normalize_csr(rowptr, val):
    """D^-1 A -- nn/utils/utils.py:240-247 (rows summing to 0 keep zeros)."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    out = np.array(val, dtype=np.float32, copy=True)
    for r in range(rowptr.shape[0] - 1):
        b, e = rowptr[r], rowptr[r + 1]
        s = out[b:e].sum(dtype=np.float32)
        if e > b and s != 0:
            out[b:e] *= np.float32(1.0) / s
    return out
