"""HIP kernels (through the C ABI) vs the CPU oracle on seeded random graphs.  GPU box only (-m gpu)."""
import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu


def np_graph(n, avg_deg, seed, heavy_rows=(), empty_rows=(), weighted=True, n_cols=None):
    """Random CSR with optional very long rows (> the 512-edge chunk threshold) and forced-empty rows."""
    rng = np.random.default_rng(seed)
    n_cols = n if n_cols is None else n_cols
    deg = rng.poisson(avg_deg, n)
    for r, d in heavy_rows:
        deg[r] = d
    for r in empty_rows:
        deg[r] = 0
    deg = np.minimum(deg, n_cols)
    rowptr = np.zeros(n + 1, np.int64)
    np.cumsum(deg, out=rowptr[1:])
    col = np.concatenate([np.sort(rng.choice(n_cols, d, replace=False)) for d in deg] + [np.zeros(0, np.int64)]).astype(np.int32)
    val = rng.standard_normal(col.shape[0]).astype(np.float32) if weighted else None
    return rowptr, col, val


def to_dev(rowptr, col, val, n_cols, device):
    import dgll_amd

    return dgll_amd.CSRGraph(torch.from_numpy(rowptr).to(device), torch.from_numpy(col).to(device),
                             None if val is None else torch.from_numpy(val).to(device), len(rowptr) - 1, n_cols)


def bf16_round(x):
    return torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()


@pytest.mark.parametrize("feat", [1, 7, 16, 47, 64, 100, 128, 256, 300, 602])
@pytest.mark.parametrize("weighted", [True, False])
def test_spmm_f32_vs_oracle(cuda_device, feat, weighted):
    from dgll_amd import ops

    n = 700
    rowptr, col, val = np_graph(n, 9, seed=feat, heavy_rows=[(3, 650), (500, 699)], empty_rows=[0, 17, n - 1], weighted=weighted)
    x = np.random.default_rng(1).standard_normal((n, feat)).astype(np.float32)
    g = to_dev(rowptr, col, val, n, cuda_device)
    assert g.num_long_rows() >= 2
    y = ops.spmm(g, torch.from_numpy(x).to(cuda_device))
    np.testing.assert_allclose(y.cpu().numpy(), cref.spmm_csr(rowptr, col, val, x), rtol=1e-4, atol=1e-4)
    ym = ops.spmm(g, torch.from_numpy(x).to(cuda_device), reduce="mean")
    np.testing.assert_allclose(ym.cpu().numpy(), cref.spmm_csr(rowptr, col, val, x, reduce="mean"), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("feat", [8, 47, 100, 128, 256, 520])
def test_spmm_bf16_vs_oracle(cuda_device, feat):
    """bf16 storage, fp32 accumulation: compare with the fp32 oracle fed the same bf16-rounded inputs; the only
    extra error is the final round-to-bf16 of the output (rel 2^-8)."""
    from dgll_amd import ops

    n = 600
    rowptr, col, val = np_graph(n, 12, seed=feat + 1, heavy_rows=[(9, 590)], empty_rows=[5])
    x = bf16_round(np.random.default_rng(2).standard_normal((n, feat)).astype(np.float32))
    g = to_dev(rowptr, col, val, n, cuda_device)
    xd = torch.from_numpy(x).to(cuda_device).to(torch.bfloat16)
    ref = cref.spmm_csr(rowptr, col, val, x)
    y = ops.spmm(g, xd)
    assert y.dtype == torch.bfloat16
    np.testing.assert_allclose(y.float().cpu().numpy(), ref, rtol=8e-3, atol=8e-3)
    y32 = ops.spmm_raw(g, xd, out_dtype=torch.float32)      # bf16 gather, fp32 output: no output rounding
    np.testing.assert_allclose(y32.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)


def test_spmm_epilogue_and_backward(cuda_device):
    from dgll_amd import ops

    n, feat = 300, 40
    rowptr, col, val = np_graph(n, 6, seed=3, heavy_rows=[(1, 290)], empty_rows=[2])
    g = to_dev(rowptr, col, val, n, cuda_device)
    rng = np.random.default_rng(4)
    x = torch.from_numpy(rng.standard_normal((n, feat)).astype(np.float32)).to(cuda_device).requires_grad_()
    bias = torch.from_numpy(rng.standard_normal(feat).astype(np.float32)).to(cuda_device).requires_grad_()
    gout = torch.from_numpy(rng.standard_normal((n, feat)).astype(np.float32)).to(cuda_device)
    y = ops.spmm(g, x, bias=bias, relu=True)
    ref_pre = cref.spmm_csr(rowptr, col, val, x.detach().cpu().numpy()) + bias.detach().cpu().numpy()
    np.testing.assert_allclose(y.detach().cpu().numpy(), np.maximum(ref_pre, 0), rtol=1e-4, atol=1e-4)
    (y * gout).sum().backward()
    gm = gout.cpu().numpy() * (ref_pre > 0)
    r, c = np.repeat(np.arange(n), np.diff(rowptr)), col
    t_rowptr, t_col, t_val = cref.coo_to_csr(c, r, val, n)
    np.testing.assert_allclose(x.grad.cpu().numpy(), cref.spmm_csr(t_rowptr, t_col, t_val, gm), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(bias.grad.cpu().numpy(), gm.sum(0), rtol=1e-4, atol=1e-3)


def test_spmm_mean_backward_and_edge_value_grad(cuda_device):
    from dgll_amd import ops

    n, feat = 200, 24
    rowptr, col, val = np_graph(n, 5, seed=8, empty_rows=[7])
    g = to_dev(rowptr, col, None, n, cuda_device)
    rng = np.random.default_rng(5)
    xn = rng.standard_normal((n, feat)).astype(np.float32)
    gn = rng.standard_normal((n, feat)).astype(np.float32)
    x = torch.from_numpy(xn).to(cuda_device).requires_grad_()
    v = torch.from_numpy(val).to(cuda_device).requires_grad_()
    y = ops.spmm(g, x, val=v, reduce="mean")
    (y * torch.from_numpy(gn).to(cuda_device)).sum().backward()
    # torch CPU autograd of the same formula as the reference oracle
    xt = torch.from_numpy(xn).requires_grad_()
    vt = torch.from_numpy(val).requires_grad_()
    row = torch.from_numpy(np.repeat(np.arange(n), np.diff(rowptr)))
    deg = torch.from_numpy(np.diff(rowptr)).clamp(min=1).float()
    yt = torch.zeros(n, feat).index_add_(0, row, vt[:, None] * xt[torch.from_numpy(col).long()]) / deg[:, None]
    (yt * torch.from_numpy(gn)).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), yt.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(x.grad.cpu().numpy(), xt.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(v.grad.cpu().numpy(), vt.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_spmm_unaligned_leading_dimension_takes_generic_path(cuda_device):
    from dgll_amd import ops

    n, feat = 150, 37
    rowptr, col, val = np_graph(n, 4, seed=11, heavy_rows=[(0, 140)])
    g = to_dev(rowptr, col, val, n, cuda_device)
    x = np.random.default_rng(6).standard_normal((n, feat)).astype(np.float32)
    big = torch.zeros(n, feat + 1, device=cuda_device)
    xd = big[:, 1:]                          # data pointer 4 bytes off 16-byte alignment, ld = 38
    xd.copy_(torch.from_numpy(x))
    out = torch.empty(n, feat, device=cuda_device)
    y = ops.spmm_raw(g, xd, out=out)
    np.testing.assert_allclose(y.cpu().numpy(), cref.spmm_csr(rowptr, col, val, x), rtol=1e-4, atol=1e-4)


def test_spmm_is_bit_reproducible_and_handles_empty_graph(cuda_device):
    import dgll_amd
    from dgll_amd import ops

    n, feat = 2000, 256
    rowptr, col, val = np_graph(n, 30, seed=12, heavy_rows=[(5, 1900), (6, 1500)])
    g = to_dev(rowptr, col, val, n, cuda_device)
    x = torch.randn(n, feat, device=cuda_device, dtype=torch.bfloat16)
    a, b = ops.spmm_raw(g, x), ops.spmm_raw(g, x)
    assert torch.equal(a, b)
    empty = dgll_amd.CSRGraph(torch.zeros(n + 1, dtype=torch.int64, device=cuda_device),
                              torch.zeros(0, dtype=torch.int32, device=cuda_device), None, n, n)
    assert float(ops.spmm_raw(empty, x).float().abs().max()) == 0.0


def test_rectangular_block_and_fixed_fanout(cuda_device):
    """Sampled-block shape: fewer destination rows than source rows; and the [N, K, D] mean of sageconv.py:33-34."""
    import dgll_amd
    from dgll_amd import ops

    n_dst, n_src, feat = 64, 500, 100
    rowptr, col, _ = np_graph(n_dst, 10, seed=13, weighted=False, n_cols=n_src)
    g = to_dev(rowptr, col, None, n_src, cuda_device)
    x = np.random.default_rng(7).standard_normal((n_src, feat)).astype(np.float32)
    y = ops.spmm(g, torch.from_numpy(x).to(cuda_device), reduce="mean")
    np.testing.assert_allclose(y.cpu().numpy(), cref.spmm_csr(rowptr, col, None, x, reduce="mean"), rtol=1e-4, atol=1e-5)
    N, K, D = 33, 25, 64
    nbr = np.random.default_rng(8).standard_normal((N, K, D)).astype(np.float32)
    fg = dgll_amd.CSRGraph.fixed_fanout(N, K, cuda_device)
    ym = ops.spmm(fg, torch.from_numpy(nbr).to(cuda_device).view(N * K, D), reduce="mean")
    np.testing.assert_allclose(ym.cpu().numpy(), nbr.mean(1), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("feat", [24, 100, 256, 700])
def test_sddmm_vs_oracle(cuda_device, dtype, feat):
    from dgll_amd import ops

    n = 300
    rowptr, col, _ = np_graph(n, 8, seed=feat, heavy_rows=[(4, 280)], empty_rows=[0], weighted=False)
    rng = np.random.default_rng(9)
    gn = rng.standard_normal((n, feat)).astype(np.float32)
    bn = rng.standard_normal((n, feat)).astype(np.float32)
    if dtype == torch.bfloat16:
        gn, bn = bf16_round(gn), bf16_round(bn)
    g = to_dev(rowptr, col, None, n, cuda_device)
    out = ops.sddmm_raw(g, torch.from_numpy(gn).to(cuda_device).to(dtype), torch.from_numpy(bn).to(cuda_device).to(dtype))
    np.testing.assert_allclose(out.cpu().numpy(), cref.sddmm_csr(rowptr, col, gn, bn), rtol=1e-4, atol=1e-4 * np.sqrt(feat))


@pytest.mark.parametrize("heads,fo", [(1, 32), (8, 8), (8, 32), (3, 5), (1, 47), (2, 128)])
@pytest.mark.parametrize("mode", [0, 1])
def test_gat_forward_vs_oracle(cuda_device, heads, fo, mode):
    from dgll_amd import ops

    n = 400
    rowptr, col, _ = np_graph(n, 7, seed=heads * 100 + fo, heavy_rows=[(2, 390)], weighted=False)
    # every row needs an edge (gatconv.py:139-141): add self loops
    dense = np.zeros((n, n), bool)
    dense[np.repeat(np.arange(n), np.diff(rowptr)), col] = True
    np.fill_diagonal(dense, True)
    r, c = np.nonzero(dense)
    rowptr, col, _ = cref.coo_to_csr(r, c, None, n)
    rng = np.random.default_rng(10)
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
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("heads,fo", [(4, 8), (1, 20)])
def test_gat_backward_vs_torch_autograd(cuda_device, mode, heads, fo):
    """Gradients of the fused kernel vs CPU autograd of the oracle's restated formula (oracle/torch_ref.py)."""
    from dgll_amd import ops
    from oracle import torch_ref

    n, fin = 250, 13
    rowptr, col, _ = np_graph(n, 6, seed=77 + heads, heavy_rows=[(1, 240)], weighted=False)
    dense = np.zeros((n, n), bool)
    dense[np.repeat(np.arange(n), np.diff(rowptr)), col] = True
    np.fill_diagonal(dense, True)
    r, c = np.nonzero(dense)
    rowptr, col, _ = cref.coo_to_csr(r, c, None, n)
    rng = np.random.default_rng(21)
    xn = rng.standard_normal((n, fin)).astype(np.float32)
    Wn = (0.3 * rng.standard_normal((heads, fin, fo))).astype(np.float32)
    an = (0.3 * rng.standard_normal((heads, 2 * fo))).astype(np.float32)
    gn = rng.standard_normal((n, heads * fo)).astype(np.float32)
    # oracle side
    xt, Wt, at = (torch.from_numpy(v).requires_grad_() for v in (xn, Wn, an))
    if mode == 0:
        yt = torch_ref.spgat_conv(torch.from_numpy(rowptr), torch.from_numpy(col), xt, Wt, at[:, None, :], 0.2, True)
    else:
        yt = torch_ref.gat_conv(torch.from_numpy(rowptr), torch.from_numpy(col), xt, Wt, at[:, :, None], 0.2, True)
    (yt * torch.from_numpy(gn)).sum().backward()
    # device side
    dev = cuda_device
    xd, Wd, ad = (torch.from_numpy(v).to(dev).requires_grad_() for v in (xn, Wn, an))
    g = to_dev(rowptr, col, None, n, dev)
    h = torch.mm(xd, torch.cat(list(Wd), dim=1))
    hv = h.view(n, heads, fo)
    s = (hv * ad[:, :fo]).sum(-1)
    t = (hv * ad[:, fo:]).sum(-1)
    fo_pad = ops.head_width_padded(fo, torch.float32)
    hp = torch.nn.functional.pad(hv, (0, fo_pad - fo)).reshape(n, heads * fo_pad)
    out = ops.gat_aggregate(g, hp, s, t, heads, 0.2, apply_elu=True, mode=mode)
    out = out.view(n, heads, fo_pad)[:, :, :fo].reshape(n, heads * fo)
    np.testing.assert_allclose(out.detach().cpu().numpy(), yt.detach().numpy(), rtol=1e-4, atol=1e-5)
    (out * torch.from_numpy(gn).to(dev)).sum().backward()
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xt.grad.numpy(), rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(Wd.grad.cpu().numpy(), Wt.grad.numpy(), rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(ad.grad.cpu().numpy(), at.grad.numpy(), rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_segment_max_vs_oracle(cuda_device, dtype):
    from dgll_amd import ops

    n, feat = 300, 50
    rowptr, col, _ = np_graph(n, 8, seed=31, heavy_rows=[(3, 280)], empty_rows=[1], weighted=False)
    x = np.random.default_rng(12).standard_normal((n, feat)).astype(np.float32)
    if dtype == torch.bfloat16:
        x = bf16_round(x)
    g = to_dev(rowptr, col, None, n, cuda_device)
    xd = torch.from_numpy(x).to(cuda_device).to(dtype).requires_grad_()
    y = ops.segment_max(g, xd)
    np.testing.assert_array_equal(y.detach().float().cpu().numpy(), cref.spmm_csr_max(rowptr, col, x))
    y.float().sum().backward()
    assert float(xd.grad.float().sum()) == float((np.diff(rowptr) > 0).sum() * feat)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["csr", "csr_with_duplicates", "block"])
def test_segment_max_backward_kernel_routes_the_gradient_to_the_argmax_row(cuda_device, dtype, kind):
    """dgll_hip_segment_max_bwd (gather over the transposed structure, no atomics) vs CPU autograd of torch's max over the
    K axis (sageconv.py:37-38): general CSR with long / empty rows, a CSR that stores some (row, col) pairs twice, and a
    sampled block (col == arange, the case of every mini-batch)."""
    import dgll_amd
    from dgll_amd import ops

    rng = np.random.default_rng(7)
    feat = 37
    if kind == "block":
        n, k = 300, 9
        g = dgll_amd.CSRGraph.fixed_fanout(n, k, cuda_device)
        rowptr, col = g.rowptr.cpu().numpy(), g.col.cpu().numpy()
        n_src = n * k
    else:
        n = n_src = 400
        rowptr, col, _ = np_graph(n, 12, seed=5, heavy_rows=[(3, 380)], empty_rows=[0, 7], weighted=False)
        if kind == "csr_with_duplicates":            # every third edge stored twice (adjacent slots of the sorted row)
            deg = np.diff(rowptr)
            rep = np.ones(col.size, dtype=np.int64)
            rep[::3] = 2
            col = np.repeat(col, rep)
            newdeg = np.add.reduceat(rep, rowptr[:-1][deg > 0]) if col.size else deg
            full = np.zeros(n, dtype=np.int64)
            full[deg > 0] = newdeg
            rowptr = np.zeros(n + 1, np.int64)
            np.cumsum(full, out=rowptr[1:])
        g = to_dev(rowptr, col, None, n_src, cuda_device)
    x = torch.from_numpy(rng.standard_normal((n_src, feat)).astype(np.float32)).to(dtype)
    xd = x.to(cuda_device).requires_grad_(True)
    w = torch.from_numpy(rng.standard_normal((len(rowptr) - 1, feat)).astype(np.float32))
    y = ops.segment_max(g, xd)
    (y.float() * w.to(cuda_device)).sum().backward()
    # CPU reference: per row, max over the listed source rows (first arg-max = lowest source id, as the kernel)
    xr = x.float().clone().requires_grad_(True)
    rows = []
    for i in range(len(rowptr) - 1):
        c = np.unique(col[rowptr[i]:rowptr[i + 1]])
        rows.append(xr[torch.from_numpy(c).long()].max(0)[0] if c.size else torch.zeros(feat))
    yr = torch.stack(rows)
    (yr * w).sum().backward()
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    torch.testing.assert_close(y.detach().float().cpu(), yr.detach(), rtol=tol, atol=tol)
    torch.testing.assert_close(xd.grad.float().cpu(), xr.grad, rtol=tol, atol=tol * float(xr.grad.abs().max()))
    # bit-reproducible: no atomics
    xd2 = x.to(cuda_device).requires_grad_(True)
    (ops.segment_max(g, xd2).float() * w.to(cuda_device)).sum().backward()
    assert torch.equal(xd2.grad, xd.grad)


@pytest.mark.parametrize("F_padded,actual_F,H", [(52, 50, 64), (64, 64, 121), (16, 13, 5), (128, 128, 32)])
def test_fused_gcn_launcher_vs_oracle(cuda_device, F_padded, actual_F, H):
    """a10: the reference-named launcher (gcn_fused_kernel.cu:190-195) vs the oracle restatement of :39-69, through both
    the explicit-stream twin and the verbatim void symbol."""
    import ctypes

    from dgll_amd import _lib, fused_gcn

    n = 500
    rowptr, col, val = np_graph(n, 10, seed=F_padded, heavy_rows=[(2, 450)], empty_rows=[3])
    rng = np.random.default_rng(33)
    X = rng.standard_normal((n, F_padded)).astype(np.float32)
    W = (rng.standard_normal((F_padded, H)) / np.sqrt(actual_F)).astype(np.float32)
    ref = cref.gcn_fused_fwd(rowptr, col, val, X, W, actual_F)
    d = cuda_device
    rp = torch.from_numpy(rowptr.astype(np.int32)).to(d)
    ci, va = torch.from_numpy(col).to(d), torch.from_numpy(val).to(d)
    Xd, Wd = torch.from_numpy(X).to(d), torch.from_numpy(W).to(d)
    nn_ = torch.from_numpy(np.diff(rowptr).astype(np.int32)).to(d)
    out = fused_gcn.gcn_fused_forward(rp, ci, va, Xd, Wd, nn_, actual_F)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    H_out = torch.zeros(n, H, device=d)
    torch.cuda.synchronize()
    _lib.lib.launch_gcn_fused_kernel(rp.data_ptr(), ci.data_ptr(), va.data_ptr(), Xd.data_ptr(), Wd.data_ptr(),
                                     H_out.data_ptr(), nn_.data_ptr(), n, F_padded, actual_F, H, int(ci.numel()))
    np.testing.assert_allclose(H_out.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    # backward: the math of relu(A.(X.W)) vs CPU autograd
    Xt, Wt = torch.from_numpy(X).requires_grad_(), torch.from_numpy(W).requires_grad_()
    row = torch.from_numpy(np.repeat(np.arange(n), np.diff(rowptr)))
    S = Xt[:, :actual_F] @ Wt[:actual_F]
    Y = torch.relu(torch.zeros(n, H).index_add_(0, row, torch.from_numpy(val)[:, None] * S[torch.from_numpy(col).long()]))
    gout = torch.from_numpy(rng.standard_normal((n, H)).astype(np.float32))
    (Y * gout).sum().backward()
    Xg, Wg = Xd.clone().requires_grad_(), Wd.clone().requires_grad_()
    Yd = fused_gcn.GCNFusedFunction.apply(rp, ci, va, Xg, Wg, nn_, actual_F)
    (Yd * gout.to(d)).sum().backward()
    np.testing.assert_allclose(Xg.grad.cpu().numpy(), Xt.grad.numpy(), rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(Wg.grad.cpu().numpy(), Wt.grad.numpy(), rtol=2e-3, atol=2e-3)
    # the reference's backward symbol (gcn_fused_kernel.cu:238-244), verbatim signature, correct math
    gW, gX, go = torch.full_like(Wd, 7.0), torch.full_like(Xd, 7.0), gout.to(d).contiguous()
    torch.cuda.synchronize()
    _lib.lib.launch_gcn_fused_kernel_backward_optimized(rp.data_ptr(), ci.data_ptr(), va.data_ptr(), Xd.data_ptr(), Wd.data_ptr(),
                                                        go.data_ptr(), gW.data_ptr(), gX.data_ptr(), nn_.data_ptr(), n, F_padded,
                                                        actual_F, H, int(ci.numel()))
    np.testing.assert_allclose(gX.cpu().numpy(), Xt.grad.numpy(), rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(gW.cpu().numpy(), Wt.grad.numpy(), rtol=2e-3, atol=2e-3)


def test_gemm_f32_kernel(cuda_device):
    from dgll_amd import _lib

    rng = np.random.default_rng(5)
    M, K, N = 333, 50, 121
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((K, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    Ad, Bd, bd = (torch.from_numpy(v).to(cuda_device) for v in (A, B, bias))
    C = torch.empty(M, N, device=cuda_device)
    code = _lib.lib.dgll_hip_gemm_f32(torch.cuda.current_stream().cuda_stream, Ad.data_ptr(), K, Bd.data_ptr(), N, C.data_ptr(),
                                      N, M, N, K, bd.data_ptr(), 1)
    assert code == 0
    np.testing.assert_allclose(C.cpu().numpy(), np.maximum(cref.gemm(A, B, bias), 0), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_spmm_accumulate_and_row_scale(cuda_device, dtype):
    """dgll_hip_spmm_csr_ex: Y = scale * (A.X + Y_old) -- the two-CSR (owned / halo columns) form of the partitioned
    path must equal one launch over the union, incl. rows longer than the chunk threshold."""
    from dgll_amd import ops

    n, feat = 400, 72
    rowptr, col, _ = np_graph(n, 10, seed=91, heavy_rows=[(7, 380)], empty_rows=[3], weighted=False)
    x = np.random.default_rng(3).standard_normal((n, feat)).astype(np.float32)
    if dtype == torch.bfloat16:
        x = bf16_round(x)
    full = to_dev(rowptr, col, None, n, cuda_device)
    xd = torch.from_numpy(x).to(cuda_device).to(dtype)
    ref = ops.spmm_raw(full, xd, reduce="mean", out_dtype=torch.float32)
    # split the edges by column parity into two CSRs over the same rows
    row = np.repeat(np.arange(n), np.diff(rowptr))
    halves = []
    for parity in (0, 1):
        m = (col % 2) == parity
        ptr = np.zeros(n + 1, np.int64)
        np.cumsum(np.bincount(row[m], minlength=n), out=ptr[1:])
        halves.append(to_dev(ptr, col[m], None, n, cuda_device))
    inv_deg = torch.from_numpy((1.0 / np.maximum(np.diff(rowptr), 1)).astype(np.float32)).to(cuda_device)
    out = torch.empty(n, feat, device=cuda_device, dtype=torch.float32)
    ops.spmm_raw(halves[0], xd, out=out)
    ops.spmm_raw(halves[1], xd, out=out, row_scale=inv_deg, accumulate=True)
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_spmm_increment_form_leaves_edgeless_rows_alone(cuda_device, dtype):
    """accumulate = 2: Y += gate(scale * A.X); rows without an edge are not read or written (they may hold anything, here
    NaN-free sentinels), chunked long rows included."""
    from dgll_amd import ops

    n, feat = 600, 136
    rowptr, col, val = np_graph(n, 3, seed=23, heavy_rows=[(9, 500), (10, 140)], empty_rows=list(range(100, 400)), weighted=True)
    g = to_dev(rowptr, col, val, n, cuda_device)
    rng = np.random.default_rng(2)
    x = torch.from_numpy(rng.standard_normal((n, feat)).astype(np.float32)).to(cuda_device).to(dtype)
    base = torch.from_numpy(rng.standard_normal((n, feat)).astype(np.float32)).to(cuda_device).to(dtype)
    gate = torch.from_numpy(rng.standard_normal((n, feat)).astype(np.float32)).to(cuda_device).to(dtype)
    scale = torch.from_numpy(rng.random(n).astype(np.float32) + 0.5).to(cuda_device)
    inc = ops.spmm_raw(g, x, row_scale=scale, out_dtype=torch.float32)                  # scale * A.X in fp32
    inc = torch.where(gate.float() > 0, inc, torch.zeros_like(inc))
    ref = (base.float() + inc).to(dtype)
    out = ops.spmm_raw(g, x, out=base.clone(), row_scale=scale, accumulate=2, gate=gate)
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=1e-2, atol=2e-2)
    torch.testing.assert_close(out.float(), ref.float(), **tol)
    empty = torch.from_numpy(np.diff(rowptr) == 0).to(cuda_device)
    assert torch.equal(out[empty], base[empty])                                          # bit-untouched
    with pytest.raises(RuntimeError):
        ops.spmm_raw(g, x, out=base.clone(), accumulate=2, relu=True)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("feat", [72, 256, 5])
def test_spmm_gate_epilogue(cuda_device, dtype, feat):
    """dgll_hip_spmm_csr_gated: Y = (A.X + Y_old) with elements zeroed where gate <= 0 -- the ReLU backward of the layer
    below fused into the kernel that produces the gradient; incl. chunked long rows, empty rows, -0.0 and negative gates."""
    from dgll_amd import ops

    n = 500
    rowptr, col, val = np_graph(n, 9, seed=17, heavy_rows=[(11, 450), (12, 140)], empty_rows=[0, 5], weighted=True)
    g = to_dev(rowptr, col, val, n, cuda_device)
    rng = np.random.default_rng(8)
    x = torch.from_numpy(rng.standard_normal((n, feat)).astype(np.float32)).to(cuda_device).to(dtype)
    gate = torch.from_numpy(rng.standard_normal((n, feat)).astype(np.float32)).to(cuda_device).to(dtype)
    gate[3] = 0.0
    gate[4] = -0.0
    base = torch.from_numpy(rng.standard_normal((n, feat)).astype(np.float32)).to(cuda_device).to(dtype)
    plain = ops.spmm_raw(g, x, out=base.clone(), accumulate=True)
    ref = torch.where(gate > 0, plain, torch.zeros_like(plain))
    out = ops.spmm_raw(g, x, out=base.clone(), accumulate=True, gate=gate)
    assert torch.equal(out, ref)
    with pytest.raises(ValueError):
        ops.spmm_raw(g, x, gate=gate[:, :-1])
    if dtype == torch.bfloat16:      # bf16 gather into an fp32 output (two 16-byte loads of the previous row / the gate per lane)
        base32, gate32 = base.float(), gate.float()
        plain32 = ops.spmm_raw(g, x, out=base32.clone(), accumulate=True)
        out32 = ops.spmm_raw(g, x, out=base32.clone(), accumulate=True, gate=gate32)
        assert torch.equal(out32, torch.where(gate32 > 0, plain32, torch.zeros_like(plain32)))


@pytest.mark.parametrize("mode", [0, 1])
def test_gat_attention_dropout_scale_forward_and_backward(cuda_device, mode):
    """edge_scale path (attention dropout, gatconv.py:37,132): multipliers applied AFTER the row sum; forward and the
    gradients vs CPU autograd of the same formula with the same mask."""
    from dgll_amd import ops

    n, heads, fo = 200, 2, 8
    rowptr, col, _ = np_graph(n, 6, seed=5, heavy_rows=[(0, 190)], weighted=False)
    dense = np.zeros((n, n), bool)
    dense[np.repeat(np.arange(n), np.diff(rowptr)), col] = True
    np.fill_diagonal(dense, True)
    r, c = np.nonzero(dense)
    rowptr, col, _ = cref.coo_to_csr(r, c, None, n)
    rng = np.random.default_rng(8)
    hn = (0.5 * rng.standard_normal((n, heads * fo))).astype(np.float32)
    sn = rng.standard_normal((n, heads)).astype(np.float32)
    tn = rng.standard_normal((n, heads)).astype(np.float32)
    scale = ((rng.random((len(col), heads)) > 0.4) / 0.6).astype(np.float32)       # dropout p = 0.4
    gn = rng.standard_normal((n, heads * fo)).astype(np.float32)
    # CPU autograd reference
    ht, st, tt = (torch.from_numpy(v).requires_grad_() for v in (hn, sn, tn))
    row = torch.from_numpy(np.repeat(np.arange(n), np.diff(rowptr)))
    colt = torch.from_numpy(col).long()
    sign = -1.0 if mode == 0 else 1.0
    z = sign * torch.nn.functional.leaky_relu(st[row] + tt[colt], 0.2)            # [E, heads]
    if mode == 1:
        mx = torch.full((n, heads), -float("inf")).scatter_reduce(0, row[:, None].expand(-1, heads), z, "amax")
        z = z - mx[row]
    w = torch.exp(z)
    den = torch.zeros(n, heads).index_add_(0, row, w)
    num = torch.zeros(n, heads, fo).index_add_(0, row, (w * torch.from_numpy(scale))[:, :, None] * ht.view(n, heads, fo)[colt])
    ref = torch.nn.functional.elu(num / den[:, :, None]).reshape(n, heads * fo)
    (ref * torch.from_numpy(gn)).sum().backward()
    # device
    d = cuda_device
    g = to_dev(rowptr, col, None, n, d)
    hd, sd, td = (torch.from_numpy(v).to(d).requires_grad_() for v in (hn, sn, tn))
    out = ops.gat_aggregate(g, hd, sd, td, heads, 0.2, apply_elu=True, mode=mode, edge_scale=torch.from_numpy(scale).to(d))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-5)
    (out * torch.from_numpy(gn).to(d)).sum().backward()
    np.testing.assert_allclose(hd.grad.cpu().numpy(), ht.grad.numpy(), rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(sd.grad.cpu().numpy(), st.grad.numpy(), rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(td.grad.cpu().numpy(), tt.grad.numpy(), rtol=2e-3, atol=2e-4)


def test_training_mode_gat_runs_with_dropout(cuda_device):
    from dgll_amd import nn as dnn

    torch.manual_seed(0)
    n = 300
    adj = (torch.rand(n, n) < 0.05).float()
    adj.fill_diagonal_(1.0)
    model = dnn.SpGAT(16, 8, 5, dropout=0.5, alpha=0.2, nheads=4).to(cuda_device).train()
    out = model(torch.randn(n, 16, device=cuda_device), adj.to(cuda_device))
    out.sum().backward()
    assert torch.isfinite(out).all() and all(torch.isfinite(p.grad).all() for p in model.parameters())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gat_split_launches_equal_the_fused_one(cuda_device, dtype):
    """The partitioned GAT's building blocks on one GPU: the adjacency split by column parity into two CSRs; raw +
    accumulate forward launches, two-part backward (rows pass accumulated over the halves, cols pass per transposed
    half) vs the single fused forward/backward."""
    from dgll_amd import ops
    from dgll_amd.ops_edge import _empty_padded, gat_bwd_cols_part, gat_bwd_rows_part, gat_fwd_part

    n, heads, fo = 500, 4, 8
    rowptr, col, _ = np_graph(n, 9, seed=123, heavy_rows=[(5, 460)], weighted=False)
    dense = np.zeros((n, n), bool)
    dense[np.repeat(np.arange(n), np.diff(rowptr)), col] = True
    np.fill_diagonal(dense, True)
    r, c = np.nonzero(dense)
    rowptr, col, _ = cref.coo_to_csr(r, c, None, n)
    d = cuda_device
    full = to_dev(rowptr, col, None, n, d)
    row = np.repeat(np.arange(n), np.diff(rowptr))
    halves = []
    for parity in (0, 1):
        m = (col % 2) == parity
        ptr = np.zeros(n + 1, np.int64)
        np.cumsum(np.bincount(row[m], minlength=n), out=ptr[1:])
        halves.append(to_dev(ptr, col[m], None, n, d))
    torch.manual_seed(3)
    h = (0.5 * torch.randn(n, heads * fo, device=d)).to(dtype).requires_grad_()
    s = torch.randn(n, heads, device=d, requires_grad=True)
    t = torch.randn(n, heads, device=d, requires_grad=True)
    go = torch.randn(n, heads * fo, device=d).to(dtype)
    ref = ops.gat_aggregate(full, h, s, t, heads, 0.2, apply_elu=True, mode=0)
    gh_ref, gs_ref, gt_ref = torch.autograd.grad(ref, (h, s, t), go)
    # forward: raw on the first half, accumulate + normalise on the second
    out = _empty_padded(n, heads * fo, dtype, d)
    rowsum = torch.empty(n, heads, device=d)
    hd, sd, td = h.detach(), s.detach().contiguous(), t.detach().contiguous()
    gat_fwd_part(halves[0], hd, sd, td, out, rowsum, heads, fo, 0.2, True, raw=True, accumulate=False)
    gat_fwd_part(halves[1], hd, sd, td, out, rowsum, heads, fo, 0.2, True, raw=False, accumulate=True)
    tol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(out.float(), ref.detach().float(), **tol)
    # backward
    dn = _empty_padded(n, heads * fo, dtype, d)
    dd = torch.empty(n, heads, device=d)
    gs = torch.empty(n, heads, device=d)
    gat_bwd_rows_part(halves[0], hd, sd, td, out, go, rowsum, dn, dd, gs, heads, fo, 0.2, True, accumulate=False)
    gat_bwd_rows_part(halves[1], hd, sd, td, out, go, rowsum, dn, dd, gs, heads, fo, 0.2, True, accumulate=True)
    gh = torch.zeros(n, heads * fo, device=d, dtype=torch.float32)
    gt = torch.zeros(n, heads, device=d)
    for half in halves:
        ht, _ = half.transpose()
        gh_part = _empty_padded(n, heads * fo, dtype, d)
        gt_part = torch.empty(n, heads, device=d)
        gat_bwd_cols_part(ht, dn, hd, td, sd, dd, gh_part, gt_part, heads, fo, 0.2)
        gh += gh_part.float()
        gt += gt_part
    gtol = dict(rtol=2e-3, atol=2e-4) if dtype == torch.float32 else dict(rtol=5e-2, atol=8e-2)
    torch.testing.assert_close(gs, gs_ref, **gtol)
    torch.testing.assert_close(gt, gt_ref, **gtol)
    torch.testing.assert_close(gh, gh_ref.float(), **gtol)
    # one launch over the whole adjacency, DECLARED the only one (accumulate = 3): dd_i from the pass's own dot products -- equal to
    # the split launches' (dd_i from the stored output row) to rounding, and what the fused backward above ran
    dn1, dd1, gs1 = _empty_padded(n, heads * fo, dtype, d), torch.empty(n, heads, device=d), torch.empty(n, heads, device=d)
    gat_bwd_rows_part(full, hd, sd, td, out, go, rowsum, dn1, dd1, gs1, heads, fo, 0.2, True, accumulate=3)
    torch.testing.assert_close(dn1.float(), dn.float(), rtol=0, atol=0)
    torch.testing.assert_close(gs1, gs_ref, **gtol)
    torch.testing.assert_close(dd1, dd, **(dict(rtol=2e-3, atol=2e-4) if dtype == torch.float32 else dict(rtol=5e-2, atol=5e-2)))
    # the SPLIT exact pass (phases 4 / 6, partial sums parked between the launches): the same dd_i and grad_s as the single exact
    # launch up to the order of the fp32 additions -- far inside what the stored-output form (above) differs by in bf16
    dn2, dd2, gs2 = _empty_padded(n, heads * fo, dtype, d), torch.empty(n, heads, device=d), torch.empty(n, heads, device=d)
    partial = torch.empty(n, 3 * heads, device=d)
    gat_bwd_rows_part(halves[0], hd, sd, td, out, go, rowsum, dn2, dd2, gs2, heads, fo, 0.2, True, accumulate=4, partial=partial)
    gat_bwd_rows_part(halves[1], hd, sd, td, out, go, rowsum, dn2, dd2, gs2, heads, fo, 0.2, True, accumulate=6, partial=partial)
    torch.testing.assert_close(dn2.float(), dn1.float(), rtol=0, atol=0)
    torch.testing.assert_close(dd2, dd1, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(gs2, gs1, rtol=1e-4, atol=2e-5)
    with pytest.raises(ValueError):
        gat_bwd_rows_part(halves[0], hd, sd, td, out, go, rowsum, dn2, dd2, gs2, heads, fo, 0.2, True, accumulate=4)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,feat", [(torch.bfloat16, 256), (torch.bfloat16, 602), (torch.float32, 100), (torch.bfloat16, 47), (torch.float32, 7)])
@pytest.mark.parametrize("reduce", ["mean", "sum"])
def test_sampled_block_backward_is_one_expand_launch(cuda_device, dtype, feat, reduce):
    """Backward of the K-axis reduction over a sampled block (col == arange: every source row belongs to one destination row,
    base_sampler.py:30-43; sageconv.py:33-36): dgll_hip_expand_rows against the definition out[k] = g[row(k)] (/ deg for the mean),
    on an ordinary block and on a block padded to static shapes (unused rows empty, the unused tail of the sources zero)."""
    from dgll_amd import ops
    from dgll_amd.graph import CSRGraph
    from dgll_amd.graphs import PaddedBlock

    dev = cuda_device
    gen = torch.Generator().manual_seed(5)
    n_rows, fan = 300, 7
    deg = torch.randint(0, fan + 1, (n_rows,), generator=gen)
    deg[3] = 0
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64)
    torch.cumsum(deg, 0, out=rowptr[1:])
    nnz = int(rowptr[-1])
    blk = CSRGraph(rowptr.to(dev), torch.arange(nnz, dtype=torch.int32, device=dev), None, n_rows, nnz, check=False)
    blk.identity_cols, blk.max_degree = True, fan
    x = ops.alloc_features(nnz, feat, dtype, dev)
    x.copy_(torch.randn(nnz, feat, generator=gen).to(dev))
    x.requires_grad_()
    y = ops.spmm(blk, x, reduce=reduce)
    g = torch.randn(n_rows, feat, generator=gen).to(dev).to(dtype)
    y.backward(g)
    row = torch.repeat_interleave(torch.arange(n_rows), deg).to(dev)
    want = g.float()[row]
    if reduce == "mean":
        want = want / deg.clamp(min=1).float().to(dev)[row].unsqueeze(1)
    tol = 2 ** -7 if dtype == torch.bfloat16 else 1e-6
    assert x.grad.shape == (nnz, feat) and float((x.grad.float() - want).abs().max()) <= tol * max(float(want.abs().max()), 1.0)
    # the same batch on a static block with room for more rows and edges
    pb = PaddedBlock.make(n_rows + 20, fan, dev, cols=nnz + 50)
    PaddedBlock.pad(pb, rowptr.to(dev), nnz)
    xp = ops.alloc_features(nnz + 50, feat, dtype, dev)
    xp.zero_()
    xp[:nnz].copy_(x.detach())
    xp.requires_grad_()
    yp = ops.spmm(pb, xp, reduce=reduce)
    assert torch.equal(yp[:n_rows].detach(), y.detach())
    gp = torch.zeros(n_rows + 20, feat, dtype=dtype, device=dev)
    gp[:n_rows] = g
    gp[n_rows:] = 7.0                                   # gradient of rows the batch does not use: reaches nobody
    yp.backward(gp)
    assert torch.equal(xp.grad[:nnz], x.grad) and float(xp.grad[nnz:].abs().max()) == 0.0
