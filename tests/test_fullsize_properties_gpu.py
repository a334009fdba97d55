"""Size-independent properties of the kernels at BASELINE's full sizes (ogbn-products shape: 2.45 M nodes, ~1.1e8 nnz,
hidden 256, bf16 and fp32) -- sizes the CPU oracle cannot finish in seconds.  GPU box only."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def products(cuda_device):
    from dgll_amd import synth

    # bench.py's default workload: exact products size, permuted ids, the engine's reordering
    g = synth.products_like_graph(cuda_device, seed=0, locality=0.9, exact=True, permute_ids=True).reorder(seed=0)[0]
    assert g.nnz == 123_718_280 and g.n_rows == 2_449_029
    g.plan()
    return g


def test_mean_of_ones_is_one_and_rows_are_exact(products, cuda_device):
    """mean-aggregating an all-ones matrix gives exactly 1 on every row with a neighbour and 0 elsewhere (bf16 exact)."""
    from dgll_amd import ops

    g = products
    ones = torch.ones(g.n_cols, 256, device=cuda_device, dtype=torch.bfloat16)
    y = ops.spmm_raw(g, ones, reduce="mean")
    has = (g.degrees() > 0).to(torch.bfloat16).unsqueeze(1)
    assert torch.equal(y, has.expand_as(y))
    # sum-aggregating ones gives the degree (exact in fp32 up to 2^24)
    d = ops.spmm_raw(g, torch.ones(g.n_cols, 8, device=cuda_device), reduce="sum")
    assert torch.equal(d[:, 0].long(), g.degrees())


def test_linearity_and_checksum_of_checksums(products, cuda_device):
    from dgll_amd import ops

    g = products
    torch.manual_seed(0)
    x1 = torch.randn(g.n_cols, 128, device=cuda_device)
    x2 = torch.randn(g.n_cols, 128, device=cuda_device)
    y1, y2, y12 = ops.spmm_raw(g, x1), ops.spmm_raw(g, x2), ops.spmm_raw(g, x1 + x2)
    assert float((y12 - (y1 + y2)).abs().max()) <= 1e-3 * float(y12.abs().max())
    # 1^T (A x) == (A^T 1)^T x : column sums through the transposed CSR (built by the stable sort) vs the forward launch
    gt, _ = g.transpose()
    indeg = ops.spmm_raw(gt, torch.ones(g.n_rows, 4, device=cuda_device))[:, 0].double()
    lhs = y1.double().sum(0)
    rhs = (indeg.unsqueeze(1) * x1.double()).sum(0)
    assert float((lhs - rhs).abs().max()) <= 1e-6 * float(rhs.abs().max()) + 1e-3
    # determinism: bitwise identical re-run
    assert torch.equal(y1, ops.spmm_raw(g, x1))


def test_backward_is_the_transpose_at_full_size(products, cuda_device):
    """<A x, g> == <x, A^T g> with the mean reduce, bf16 storage."""
    from dgll_amd import ops

    g = products
    torch.manual_seed(1)
    x = torch.randn(g.n_cols, 256, device=cuda_device).to(torch.bfloat16).requires_grad_()
    go = torch.randn(g.n_rows, 256, device=cuda_device).to(torch.bfloat16)
    y = ops.spmm(g, x, reduce="mean")
    y.backward(go)
    lhs = (y.detach().double() * go.double()).sum()
    rhs = (x.detach().double() * x.grad.double()).sum()
    assert abs(float(lhs - rhs)) <= 2e-2 * abs(float(lhs)) + 1.0


def test_gat_attention_is_a_convex_combination(products, cuda_device):
    """Aggregating a per-head constant through the fused edge-softmax returns that constant (weights sum to 1),
    8 heads x 32, at products size, both sign conventions."""
    from dgll_amd import ops, synth

    g = synth.products_like_graph(cuda_device, seed=0, locality=0.9, self_loops=True)   # every row needs an edge
    heads, fo = 8, 32
    const = torch.arange(1, heads + 1, device=cuda_device, dtype=torch.float32).repeat_interleave(fo)
    h = const.unsqueeze(0).expand(g.n_cols, -1).contiguous().to(torch.bfloat16)
    torch.manual_seed(2)
    s = torch.randn(g.n_rows, heads, device=cuda_device)
    t = torch.randn(g.n_cols, heads, device=cuda_device)
    for mode in (0, 1):
        out = ops.gat_aggregate(g, h, s, t, heads, 0.2, apply_elu=False, mode=mode)
        err = (out.float() - const.unsqueeze(0)).abs().max()
        assert float(err) <= 0.04 * heads, (mode, float(err))      # bf16 output rounding of values up to 8


@pytest.mark.parametrize("mode", [0, 1])
def test_gat_backward_passes_at_full_size(cuda_device, mode):
    """BASELINE config 4 shape (8 heads x 32, products size): both backward gather passes of the fused edge-softmax.
    (1) For fixed scores the layer is linear in h, so <P h, g> == <h, P^T g> checks the column pass (grad_h);
    (2) the row pass (grad_s) and the score part of the column pass (grad_t) against a central difference of the forward
        along a random direction.  fp32 features so the difference quotient is not drowned in bf16 rounding."""
    from dgll_amd import ops, synth

    g = synth.products_like_graph(cuda_device, seed=0, locality=0.9, self_loops=True, exact=True)   # every row needs an edge
    heads, fo = 8, 32
    torch.manual_seed(3)
    h = torch.randn(g.n_cols, heads * fo, device=cuda_device).requires_grad_()
    s = (0.5 * torch.randn(g.n_rows, heads, device=cuda_device)).requires_grad_()
    t = (0.5 * torch.randn(g.n_cols, heads, device=cuda_device)).requires_grad_()
    go = torch.randn(g.n_rows, heads * fo, device=cuda_device)

    def f(hh, ss, tt):
        return ops.gat_aggregate(g, hh, ss, tt, heads, 0.2, apply_elu=False, mode=mode)

    out = f(h, s, t)
    assert torch.isfinite(out).all()
    out.backward(go)
    lhs = (out.detach().double() * go.double()).sum()
    rhs = (h.detach().double() * h.grad.double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-4 * abs(float(lhs)) + 1e-2, (float(lhs), float(rhs))
    ds, dt = torch.randn_like(s), torch.randn_like(t)
    eps = 1e-2
    with torch.no_grad():
        fp = (f(h.detach(), s + eps * ds, t + eps * dt).double() * go.double()).sum()
        fm = (f(h.detach(), s - eps * ds, t - eps * dt).double() * go.double()).sum()
    numeric = float(fp - fm) / (2 * eps)
    analytic = float((s.grad.double() * ds.double()).sum() + (t.grad.double() * dt.double()).sum())
    scale = float(s.grad.double().norm() * ds.double().norm() + t.grad.double().norm() * dt.double().norm())
    assert abs(numeric - analytic) <= 2e-2 * abs(analytic) + 1e-4 * scale, (numeric, analytic)
    # bit-reproducible (no atomics in either pass)
    h2 = h.detach().clone().requires_grad_()
    s2, t2 = s.detach().clone().requires_grad_(), t.detach().clone().requires_grad_()
    f(h2, s2, t2).backward(go)
    assert torch.equal(h2.grad, h.grad) and torch.equal(s2.grad, s.grad) and torch.equal(t2.grad, t.grad)


def test_rmat27_int64_rowptr_path_is_exact(cuda_device):
    """BASELINE config 5 shape on one GPU: RMAT-27, 134 M nodes, 2.26e9 nonzeros (> 2^31: the int64 row-pointer path for
    real), F = 128 bf16.  mean-aggregating ones is exactly 1 on every row; sum-aggregating ones is the degree."""
    from dgll_amd import ops, synth

    g = synth.rmat_graph(27, 16, seed=0, device=cuda_device, symmetric=False, weighted=False, self_loops=True)
    assert g.nnz > 2 ** 31 and g.n_rows == 1 << 27
    torch.cuda.empty_cache()
    ones = torch.ones(g.n_cols, 128, device=cuda_device, dtype=torch.bfloat16)
    y = ops.spmm_raw(g, ones, reduce="mean")
    assert bool((y == 1).all())
    del y, ones
    deg = ops.spmm_raw(g, torch.ones(g.n_cols, 4, device=cuda_device), reduce="sum")[:, 0]
    small = g.degrees() < (1 << 24)                       # fp32 counts are exact below 2^24
    assert torch.equal(deg[small].long(), g.degrees()[small])
    del g, deg
    torch.cuda.empty_cache()


def test_rmat27_sized_transform_is_exact_on_integers(cuda_device):
    """BASELINE config 5's dense half, S = X.W (gcnconv.py:30) at 134 217 728 rows x 128 bf16 columns on one GPU: with
    entries in {-1, 0, 1} every product and every partial sum is an integer of magnitude <= 128 -- exact in fp32 and in the
    bf16 output -- so the MFMA kernel must reproduce an fp32 product bit for bit on any slice of rows."""
    from dgll_amd import dense

    dev = cuda_device
    m = 1 << 27
    torch.manual_seed(27)
    x = torch.randint(-1, 2, (m, 128), device=dev, dtype=torch.int8).to(torch.bfloat16)
    w = torch.randint(-1, 2, (128, 128), device=dev, dtype=torch.int8).to(torch.bfloat16)        # stored [out, in]
    y = dense.transform_bf16(x, w)
    assert y.shape == (m, 128) and y.dtype == torch.bfloat16
    for lo, hi in ((0, 4096), (m // 3, m // 3 + 4096), (m - 4096, m), ((1 << 26) - 2048, (1 << 26) + 2048)):
        ref = (x[lo:hi].float() @ w.float().t()).to(torch.bfloat16)
        assert torch.equal(y[lo:hi], ref)
    # a checksum over ALL rows: column sums of y against column sums of x pushed through w (integers, exact in fp64)
    ysum = y.double().sum(0)
    xsum = x.double().sum(0)
    assert torch.equal(ysum, xsum @ w.double().t())
    del x, y
    torch.cuda.empty_cache()


def test_dense_backward_products_at_full_size(cuda_device):
    """The MFMA weight-gradient and dual input-gradient kernels at the products row count (2 449 029 rows, 256 columns):
    exact integer cases, additivity over a row split, the adjoint identity <g.W^T, x> = <g, x.W> that ties the two kernels
    together, bit-identical reruns."""
    from dgll_amd import dense

    dev = cuda_device
    m = 2_449_029
    torch.manual_seed(1)
    # small integers: every product and every partial sum is exact in bf16 x bf16 -> fp32 (|sum| < 2^24)
    x = torch.randint(-2, 3, (m, 256), device=dev).to(torch.bfloat16)
    g = torch.randint(-2, 3, (m, 256), device=dev).to(torch.bfloat16)
    d1, d2 = dense.grad_weight_pair(x, g, g)                       # x^T.g and g^T.g
    assert torch.equal(dense.grad_weight_pair(x, g, g)[0], d1)     # rerun: bit-identical
    assert torch.equal(d2, d2.t())                                 # a Gram matrix is symmetric -- exactly, on integers
    # additivity over a row split (exact on integers, whatever the slab boundaries)
    h = m // 2 + 7
    top, bot = dense.grad_weight(x[:h], g[:h]), dense.grad_weight(x[h:], g[h:])
    assert torch.equal(top + bot, d1)
    # the shared-operand form of the narrowing layer (x^T.g1, x^T.g2 from one launch, stored transposed): on integers exactly the
    # column blocks of the full product
    n1 = 47
    a, b = dense.grad_weight_shared_x(x, g[:, :n1], g[:, 64:64 + n1])
    assert torch.equal(a, d1[:, :n1]) and torch.equal(b, d1[:, 64:64 + n1])
    # column sums through a ones operand, against an integer reduction
    ones = torch.ones(m, 64, device=dev, dtype=torch.bfloat16)
    s = dense.grad_weight(ones, g)
    assert torch.equal(s[0].long(), g.long().sum(0)) and torch.equal(s[0], s[63])
    # adjoint identity between the input-gradient and the weight-gradient kernels, <g.W^T, y> = <y^T.g, W>, taken at
    # y = the input-gradient kernel's own output so that both sides are a large positive number (a random y makes them a
    # cancelling sum of 6e8 terms whose bf16 rounding noise is as large as the value)
    gr = torch.randn(m, 256, device=dev).to(torch.bfloat16)
    w = (torch.randn(256, 256, device=dev) / 16).to(torch.bfloat16)              # stored [in, out]
    gx, gx2 = dense.transform_bf16_dual(gr, w, w)                                # g . W^T, bf16, twice
    assert torch.equal(gx, gx2)
    lhs = float((gx.double() ** 2).sum())                                        # <round(g.W^T), gx>
    rhs = float((dense.grad_weight(gx, gr).double() * w.double()).sum())         # <gx^T.g, W> = <g.W^T, gx>
    assert abs(lhs - rhs) <= 1e-3 * abs(rhs), (lhs, rhs)


def test_gate_bits_at_full_size(cuda_device):
    """The bit form of the ReLU gate at the bench shape (M = 2 449 029 rows -- not a multiple of the kernels' 256 / 512-row
    blocks --, 256 columns): the sign bits a forward transform writes are exactly the signs of what it stored (every word, popcount
    as a checksum), and the gated transform gives bit for bit the same matrix whether it reads those bits or the activation."""
    from dgll_amd import dense, ops

    M = 2_449_029
    torch.manual_seed(0)
    h = ops.alloc_features(M, 256, torch.bfloat16, cuda_device)
    h.copy_(torch.randn(M, 256, device=cuda_device))
    w = (torch.randn(256, 256, device=cuda_device) * 0.06).to(torch.bfloat16)
    out, bits = dense.transform_bf16(h, w, relu=True, bits_out=True)
    assert bits.shape == (M, 8)
    pos = out > 0
    # word by word on a sample of rows spread over the matrix (first / last block included), popcount over all of it
    rows = torch.cat([torch.arange(0, 600, device=cuda_device), torch.arange(M - 600, M, device=cuda_device),
                      torch.randint(0, M, (20_000,), device=cuda_device)])
    weights = (1 << torch.arange(32, device=cuda_device, dtype=torch.int64))
    words = (pos[rows].view(-1, 8, 32).to(torch.int64) * weights).sum(-1)
    got = bits[rows].to(torch.int64) & 0xFFFFFFFF
    assert torch.equal(got, words)
    lut = torch.tensor([bin(i).count("1") for i in range(256)], device=cuda_device, dtype=torch.int64)
    by = bits.view(torch.uint8).to(torch.int64)
    assert int(lut[by].sum()) == int(pos.sum())
    del by, words, got
    g = ops.alloc_features(M, 64, torch.bfloat16, cuda_device)
    g.copy_(torch.randn(M, 64, device=cuda_device))
    wg = (torch.randn(256, 64, device=cuda_device) * 0.1).to(torch.bfloat16)
    a = dense.transform_bf16(g, wg, out_gate=out)
    b = dense.transform_bf16(g, wg, out_gate=out, gate_bits=bits)
    assert torch.equal(a, b)
    assert bool((b[~pos] == 0).all())
