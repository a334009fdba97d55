"""Round-2 pins: row a10 (fused relu(A.X.W)), row a11 on the device, and the pooling row.

a10's own source is CUDA (dgll/FusedKernel/gcn_fused_kernel.cu:5-74) and cannot be built here, but its arithmetic --
relu(A.(X.W)), unit edge values, duplicates summed, no bias -- is exactly the RUNNABLE reference layer
Evaluation/PPI/gcn_model.py:63-77.  tests/golden/fused_gcn_layer_*.npz hold that layer's output and gradients (one
RMAT-shaped graph with duplicated edges, one graph of the bundled PPI data); the C oracle row, the reference-named
launchers and the Python face are all checked against them.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden

FUSED = ["fused_gcn_layer_n300", "fused_gcn_layer_ppi_test1"]


def _csr_with_duplicates(g):
    """int32 CSR the way the launcher takes it: every edge of the (un-coalesced) edge list is a unit-valued entry --
    torch.sparse.mm sums duplicates (gcn_model.py:56,76), a CSR with repeated columns sums them just the same."""
    n = g.meta["n"]
    ei = g["edge_index"].astype(np.int64)
    order = np.lexsort((ei[1], ei[0]))
    row, col = ei[0][order], ei[1][order]
    rowptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(row, minlength=n), out=rowptr[1:])
    return rowptr, col.astype(np.int32), np.ones(col.size, dtype=np.float32)


@pytest.mark.parametrize("name", FUSED)
def test_oracle_fused_row_matches_the_runnable_reference_layer(name):
    """Pins oracle/oracle.c's a10 row (restated from gcn_fused_kernel.cu:39-69) to vectors the reference produced."""
    from oracle import cref

    g = load_golden(name)
    rowptr, col, val = _csr_with_duplicates(g)
    F = g.meta["F"]
    Fp = (F + 3) // 4 * 4                                            # train_gcn.py:48,53-54: features padded to x4
    X = np.zeros((g.meta["n"], Fp), dtype=np.float32)
    X[:, :F] = g["x"]
    W = np.zeros((Fp, g.meta["H"]), dtype=np.float32)
    W[:F] = g["w"]
    W[F:] = 7.0                                                      # rows beyond actual_F must be ignored
    h = cref.gcn_fused_fwd(rowptr, col, val, X, W, F)
    np.testing.assert_allclose(h, g["h"], rtol=1e-4, atol=1e-4 * float(np.abs(g["h"]).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("name", FUSED)
def test_fused_launchers_match_the_runnable_reference_layer(cuda_device, name):
    """launch_gcn_fused_kernel / launch_gcn_fused_kernel_backward_optimized (reference symbol names and signatures,
    gcn_fused_kernel.cu:190-195,238-244) and the Python face (gcn_extension.cpp:103-110) vs the stored H, grad_X, grad_W."""
    from dgll_amd import _lib, fused_gcn

    g = load_golden(name)
    d = cuda_device
    n, F, H = g.meta["n"], g.meta["F"], g.meta["H"]
    rowptr, col, val = _csr_with_duplicates(g)
    Fp = (F + 3) // 4 * 4
    X = torch.zeros(n, Fp)
    X[:, :F] = g.t("x")
    W = torch.zeros(Fp, H)
    W[:F] = g.t("w")
    rp, ci, va = (torch.from_numpy(a).to(d) for a in (rowptr, col, val))
    Xd, Wd = X.to(d), W.to(d)
    nn_ = torch.from_numpy(np.diff(rowptr).astype(np.int32)).to(d)
    scale = float(np.abs(g["h"]).max())
    out = torch.zeros(n, H, device=d)
    torch.cuda.synchronize()
    _lib.lib.launch_gcn_fused_kernel(rp.data_ptr(), ci.data_ptr(), va.data_ptr(), Xd.data_ptr(), Wd.data_ptr(), out.data_ptr(),
                                     nn_.data_ptr(), n, Fp, F, H, int(ci.numel()))
    np.testing.assert_allclose(out.cpu().numpy(), g["h"], rtol=1e-4, atol=1e-4 * scale)
    np.testing.assert_allclose(fused_gcn.gcn_fused_forward(rp, ci, va, Xd, Wd, nn_, F).cpu().numpy(), g["h"], rtol=1e-4, atol=1e-4 * scale)
    gW, gX, go = torch.full_like(Wd, 3.0), torch.full_like(Xd, 3.0), g.t("gout").to(d).contiguous()
    torch.cuda.synchronize()
    _lib.lib.launch_gcn_fused_kernel_backward_optimized(rp.data_ptr(), ci.data_ptr(), va.data_ptr(), Xd.data_ptr(), Wd.data_ptr(),
                                                        go.data_ptr(), gW.data_ptr(), gX.data_ptr(), nn_.data_ptr(), n, Fp, F, H,
                                                        int(ci.numel()))
    gxs, gws = float(np.abs(g["grad_x"]).max()), float(np.abs(g["grad_w"]).max())
    np.testing.assert_allclose(gX[:, :F].cpu().numpy(), g["grad_x"], rtol=2e-3, atol=2e-4 * gxs)
    np.testing.assert_allclose(gW[:F].cpu().numpy(), g["grad_w"], rtol=2e-3, atol=2e-4 * gws)
    assert float(gX[:, F:].abs().max()) == 0.0 and float(gW[F:].abs().max()) == 0.0      # padding receives no gradient
    Xg, Wg = Xd.clone().requires_grad_(), Wd.clone().requires_grad_()
    (fused_gcn.GCNFusedFunction.apply(rp, ci, va, Xg, Wg, nn_, F) * go).sum().backward()
    np.testing.assert_allclose(Xg.grad[:, :F].cpu().numpy(), g["grad_x"], rtol=2e-3, atol=2e-4 * gxs)
    np.testing.assert_allclose(Wg.grad[:F].cpu().numpy(), g["grad_w"], rtol=2e-3, atol=2e-4 * gws)


@pytest.mark.gpu
def test_normalized_adjacency_on_the_device_matches_reference(cuda_device):
    """a11 with the edge list on the GPU: symmetrise, +I, D^-1(A+I) (nn/utils/utils.py:163-171,240-257) computed by
    dgll_amd.prep on the device == the reference's scipy pipeline (golden adj_prep_s7), structure bit-equal."""
    from dgll_amd import prep
    from dgll_amd.graph import CSRGraph

    g = load_golden("adj_prep_s7")
    n = g.meta["n"]
    src, dst = g.t("src", cuda_device), g.t("dst", cuda_device)
    key = torch.unique(src * n + dst)
    graph = prep.normalized_adjacency(torch.div(key, n, rounding_mode="floor"), key % n, n)
    assert graph.is_cuda
    ref = CSRGraph.from_coo(g.t("adj_row"), g.t("adj_col"), g.t("adj_val"), (n, n))
    assert torch.equal(graph.rowptr.cpu(), ref.rowptr) and torch.equal(graph.col.cpu(), ref.col)
    np.testing.assert_allclose(graph.val.cpu().numpy(), ref.val.numpy(), rtol=1e-6)
    coo = prep.to_torch_coo(graph)
    assert coo.is_cuda and torch.equal(coo._indices()[0].cpu(), g.t("adj_row")) and torch.equal(coo._indices()[1].cpu(), g.t("adj_col"))
    # and the SpMM consumes it: D^-1(A+I) . 1 == 1 on every row
    from dgll_amd import ops

    ones = torch.ones(n, 8, device=cuda_device)
    torch.testing.assert_close(ops.spmm_raw(graph, ones), ones, rtol=1e-6, atol=1e-6)


def _pool_golden():
    g = load_golden("pooling_scatter_reduce_b8")
    return g, g.t("x"), g.t("batch"), g.meta["B"]


def test_oracle_pooling_matches_the_scatter_reduce_vectors():
    """oracle/torch_ref.scatter_pool (restated torch_scatter semantics) vs vectors from ATen's scatter_reduce, an
    implementation independent of it (the fixture's metadata says it stands in for the absent torch_scatter)."""
    from oracle import torch_ref

    g, x, batch, B = _pool_golden()
    assert "torch_scatter" in g.meta["stands_in_for"]
    for r in ("sum", "mean", "max"):
        xr = x.clone().requires_grad_(True)
        y = torch_ref.scatter_pool(xr, batch, B, r)
        torch.testing.assert_close(y, g.t("y_" + r), rtol=1e-5, atol=1e-5)
        (y * g.t("gout")).sum().backward()
        torch.testing.assert_close(xr.grad, g.t("grad_" + r), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("scrambled", [False, True])
def test_pooling_kernels_match_the_scatter_reduce_vectors(cuda_device, scrambled):
    from dgll_amd.nn.GlobalPooling import maxPooling, meanPooling, sumPooling

    g, x, batch, B = _pool_golden()
    order = g.t("perm") if scrambled else torch.arange(x.shape[0])
    for fn, r in ((sumPooling, "sum"), (meanPooling, "mean"), (maxPooling, "max")):
        xd = x[order].to(cuda_device).requires_grad_(True)
        y = fn(xd, batch[order].to(cuda_device), B)
        torch.testing.assert_close(y.cpu(), g.t("y_" + r), rtol=1e-4, atol=1e-5)
        (y * g.t("gout", cuda_device)).sum().backward()
        expect = g.t("grad_" + r)[order]
        torch.testing.assert_close(xd.grad.cpu(), expect, rtol=1e-4, atol=1e-6)
