"""Seeded sweep over random shapes: kernels vs the CPU oracle (catches alignment / tail / tiling corner cases)."""
import numpy as np
import pytest
import torch

from oracle import cref
from test_ops_gpu import bf16_round, np_graph, to_dev

pytestmark = pytest.mark.gpu


def test_spmm_random_shape_sweep(cuda_device):
    from dgll_amd import ops

    rng = np.random.default_rng(2024)
    for case in range(40):
        n_rows = int(rng.integers(1, 900))
        n_cols = int(rng.integers(1, 900))
        feat = int(rng.choice([1, 2, 3, 5, 8, 13, 31, 32, 33, 64, 65, 96, 127, 128, 200, 255, 256, 257, 511, 513, 700]))
        avg = float(rng.choice([0.3, 2, 8, 40]))
        heavy = [(int(rng.integers(0, n_rows)), int(rng.integers(130, 600)))] if rng.random() < 0.5 else []
        weighted = bool(rng.random() < 0.5)
        bf16 = bool(rng.random() < 0.5)
        reduce = str(rng.choice(["sum", "mean"]))
        rowptr, col, val = np_graph(n_rows, avg, seed=1000 + case, heavy_rows=heavy, weighted=weighted, n_cols=n_cols)
        x = rng.standard_normal((n_cols, feat)).astype(np.float32)
        if bf16:
            x = bf16_round(x)
        g = to_dev(rowptr, col, val, n_cols, cuda_device)
        xd = torch.from_numpy(x).to(cuda_device)
        if bf16:
            xd = xd.to(torch.bfloat16)
        if rng.random() < 0.3:      # an oddly strided view: exercises the any-alignment variant
            big = torch.zeros(n_cols, feat + 3, device=cuda_device, dtype=xd.dtype)
            big[:, 1:1 + feat] = xd
            xd = big[:, 1:1 + feat]
        y = ops.spmm_raw(g, xd, reduce=reduce, out_dtype=torch.float32)
        ref = cref.spmm_csr(rowptr, col, val, x, reduce=reduce)
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=1e-4, atol=1e-5 * scale,
                                   err_msg="case %d rows %d cols %d feat %d avg %.1f weighted %s bf16 %s %s" % (
                                       case, n_rows, n_cols, feat, avg, weighted, bf16, reduce))


def test_gat_and_sddmm_random_shape_sweep(cuda_device):
    from dgll_amd import ops

    rng = np.random.default_rng(77)
    for case in range(16):
        n = int(rng.integers(2, 500))
        heads = int(rng.choice([1, 2, 3, 4, 8]))
        fo = int(rng.choice([1, 3, 4, 8, 16, 20, 32, 64]))
        mode = int(rng.integers(0, 2))
        rowptr, col, _ = np_graph(n, float(rng.choice([1, 5, 30])), seed=500 + case, weighted=False,
                                  heavy_rows=[(0, min(n, 300))] if n > 150 else [])
        dense = np.zeros((n, n), bool)
        dense[np.repeat(np.arange(n), np.diff(rowptr)), col] = True
        np.fill_diagonal(dense, True)
        r, c = np.nonzero(dense)
        rowptr, col, _ = cref.coo_to_csr(r, c, None, n)
        h = (0.5 * rng.standard_normal((n, heads * fo))).astype(np.float32)
        s = rng.standard_normal((n, heads)).astype(np.float32)
        t = rng.standard_normal((n, heads)).astype(np.float32)
        ref = cref.gat_fwd(rowptr, col, h, s, t, heads, 0.2, apply_elu=True, mode=mode)
        g = to_dev(rowptr, col, None, n, cuda_device)
        fo_pad = ops.head_width_padded(fo, torch.float32)
        hp = np.zeros((n, heads, fo_pad), np.float32)
        hp[:, :, :fo] = h.reshape(n, heads, fo)
        out = ops.gat_aggregate(g, torch.from_numpy(hp.reshape(n, -1)).to(cuda_device), torch.from_numpy(s).to(cuda_device),
                                torch.from_numpy(t).to(cuda_device), heads, 0.2, apply_elu=True, mode=mode)
        out = out.cpu().numpy().reshape(n, heads, fo_pad)[:, :, :fo].reshape(n, heads * fo)
        np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-5, err_msg="case %d n %d heads %d fo %d mode %d" % (case, n, heads, fo, mode))
        # SDDMM on the same structure
        feat = heads * fo
        gm = rng.standard_normal((n, feat)).astype(np.float32)
        e = ops.sddmm_raw(g, torch.from_numpy(gm).to(cuda_device), torch.from_numpy(h).to(cuda_device))
        np.testing.assert_allclose(e.cpu().numpy(), cref.sddmm_csr(rowptr, col, gm, h), rtol=1e-4, atol=1e-4 * np.sqrt(feat))
