"""dW = X^T . G on the split-K MFMA kernel (csrc/gradw.hip) against an fp32 CPU product of the same bf16 inputs
(sageconv.py:41,72-75 / gcnconv.py:30 leave these products to autograd)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(x, g):
    return x.double().cpu().t() @ g.double().cpu()


@pytest.mark.parametrize("m,k1,k2,n", [(20000, 256, 256, 256), (16391, 100, 100, 256), (17000, 256, 0, 47), (16384, 64, 0, 64),
                                       (33333, 47, 256, 130), (16500, 2, 254, 2)])
def test_grad_weight_matches_fp64(m, k1, k2, n):
    from dgll_amd import dense, ops

    dev = torch.device("cuda:0")
    torch.manual_seed(m + k1 + n)

    def feats(rows, cols):   # padded leading dimension with NaN in the padding: must never reach a kept output
        t = ops.alloc_features(rows, cols, torch.bfloat16, dev)
        base = t.as_strided((rows, t.stride(0)), (t.stride(0), 1))
        base.fill_(float("nan"))
        t.copy_(torch.randn(rows, cols, device=dev))
        return t

    x1, g = feats(m, k1), feats(m, n)
    x2 = feats(m, k2) if k2 else None
    assert dense._gradw_ok(x1, g)
    d1, d2 = dense._grad_weight_hip(x1, x2, g)
    r1 = _ref(x1, g)
    tol = 2e-3 * float(r1.abs().max()) + 1e-3
    assert d1.shape == (k1, n) and float((d1.double().cpu() - r1).abs().max()) <= tol
    if k2:
        r2 = _ref(x2, g)
        assert d2.shape == (k2, n) and float((d2.double().cpu() - r2).abs().max()) <= 2e-3 * float(r2.abs().max()) + 1e-3


def test_grad_weight_deterministic_and_linear():
    from dgll_amd import dense

    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    m = 50000
    x = torch.randn(m, 128, device=dev).to(torch.bfloat16)
    g = torch.randn(m, 96, device=dev).to(torch.bfloat16)
    a = dense._grad_weight_hip(x, None, g)[0]
    b = dense._grad_weight_hip(x, None, g)[0]
    assert torch.equal(a, b)                                       # fixed slab order: bit-identical reruns
    ones = torch.ones(m, 128, device=dev, dtype=torch.bfloat16)
    s = dense._grad_weight_hip(ones, None, g)[0]                   # ones^T . g = column sums of g
    ref = g.double().sum(0).cpu()
    assert float((s[0].double().cpu() - ref).abs().max()) <= 1e-3 * float(ref.abs().max()) + 1e-2
    assert torch.equal(s[0], s[77])


def test_grad_weight_used_by_layer_backward():
    """The public path: grad_weight / grad_weight_pair pick the HIP kernel for tall bf16 operands."""
    from dgll_amd import dense

    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    m = 70000
    h = torch.randn(m, 100, device=dev).to(torch.bfloat16)
    agg = torch.randn(m, 100, device=dev).to(torch.bfloat16)
    g = torch.randn(m, 256, device=dev).to(torch.bfloat16)
    gws, gwn = dense.grad_weight_pair(h, agg, g)
    for got, x in ((gws, h), (gwn, agg)):
        ref = _ref(x, g)
        assert float((got.double().cpu() - ref).abs().max()) <= 2e-3 * float(ref.abs().max())
    one = dense.grad_weight(h, g)
    assert torch.equal(one, gws)                                   # same slab order with or without the second operand? (same slabs)


@pytest.mark.parametrize("m,k,n", [(20000, 256, 256), (16390, 47, 256), (17001, 256, 100), (16384, 130, 64)])
def test_transform_dual_matches_two_products(m, k, n):
    """out1 = a.W1^T, out2 = a.W2^T from one launch == the single-product kernel run twice, and an fp64 CPU product
    within bf16 rounding."""
    from dgll_amd import dense, ops

    dev = torch.device("cuda:0")
    torch.manual_seed(m + k)
    a = ops.alloc_features(m, k, torch.bfloat16, dev)
    a.copy_(torch.randn(m, k, device=dev))
    w1 = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    w2 = (torch.randn(n, k, device=dev) / k ** 0.5).to(torch.bfloat16)
    o1, o2 = dense.transform_bf16_dual(a, w1, w2)
    for o, w in ((o1, w1), (o2, w2)):
        single = dense.transform_bf16(a, w)
        if n > 128:      # same tiling as the single-product launch: the same additions in the same order
            assert torch.equal(o, single)
        else:            # narrower outputs run the single product with other row blocks, whose reductions start at another
            d = (o.float() - single.float()).abs()      # chunk: equal up to the rounding of the last fp32 additions
            assert float(d.max()) <= 2.0 ** -6 * float(single.float().abs().max())
        ref = a.double().cpu() @ w.double().cpu().t()
        assert float((o.double().cpu() - ref).abs().max()) <= 1e-2 * float(ref.abs().max()) + 1e-3


def test_grad_weight_edge_cases():
    """Empty and tiny reductions, more slabs than rows, and the argument checks of the C entry point."""
    from dgll_amd import _lib, dense

    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    for m in (1, 15, 17, 100):                       # fewer rows than one 32-row step / than the slab count
        x = torch.randn(m, 64, device=dev).to(torch.bfloat16)
        g = torch.randn(m, 72, device=dev).to(torch.bfloat16)
        got = dense._grad_weight_hip(x, None, g)[0]
        assert float((got.double().cpu() - _ref(x, g)).abs().max()) <= 1e-3 * max(1.0, float(_ref(x, g).abs().max()))
    x0 = torch.zeros(0, 64, device=dev, dtype=torch.bfloat16)
    g0 = torch.zeros(0, 72, device=dev, dtype=torch.bfloat16)
    z = dense._grad_weight_hip(x0, None, g0)[0]
    assert z.shape == (64, 72) and float(z.abs().max()) == 0.0
    # a leading dimension that is not a multiple of 8 elements is refused, not mis-read
    x = torch.randn(64, 60, device=dev).to(torch.bfloat16)
    g = torch.randn(64, 64, device=dev).to(torch.bfloat16)
    with pytest.raises(_lib.DgllHipError):
        dense._grad_weight_hip(x, None, g)
    assert not dense._gradw_ok(x)                     # ... and the public wrappers never send it there
    ref = _ref(x, g)
    assert float((dense.grad_weight(x, g).double().cpu() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


def test_transform_dual_argument_checks():
    from dgll_amd import dense

    dev = torch.device("cuda:0")
    a = torch.randn(128, 64, device=dev).to(torch.bfloat16)
    with pytest.raises(ValueError):
        dense.transform_bf16_dual(a, torch.randn(32, 64, device=dev), torch.randn(48, 64, device=dev))
    o1, o2 = dense.transform_bf16_dual(a, torch.randn(40, 64, device=dev), torch.randn(40, 64, device=dev))
    assert o1.shape == (128, 40) and o2.shape == (128, 40)


@pytest.mark.parametrize("m,k,n1,n2", [(40000, 256, 47, 47), (16391, 100, 64, 7), (0, 256, 47, 47), (33, 130, 2, 256)])
def test_grad_weight_shared_x_is_the_two_products(m, k, n1, n2):
    """dgll_hip_grad_weight_bf16_tr: x^T . g1 and x^T . g2 from one launch (the narrowing SAGE layer's pair: the wide operand read
    once) -- bit-identical to the transposes of the untransposed launch with the roles swapped, close to the fp64 products,
    written straight into caller-provided [K, N] destinations, NaN row padding never read into a kept output."""
    from dgll_amd import dense, ops

    dev = torch.device("cuda:0")
    torch.manual_seed(m + k + n1)

    def feats(rows, cols):
        t = ops.alloc_features(rows, cols, torch.bfloat16, dev, pad_to=64)
        t.as_strided((rows, t.stride(0)), (t.stride(0), 1)).fill_(float("nan"))
        t.copy_(torch.randn(rows, cols, device=dev))
        return t

    x, g1, g2 = feats(m, k), feats(m, n1), feats(m, n2)
    o1 = torch.full((k, n1), float("nan"), device=dev)
    o2 = torch.full((k, n2), float("nan"), device=dev)
    d1, d2 = dense.grad_weight_shared_x(x, g1, g2, out1=o1, out2=o2)
    assert d1 is o1 and d2 is o2
    t1, t2 = dense._grad_weight_hip(g1, g2, x)                       # [n, k] each
    assert torch.equal(d1, t1.t()) and torch.equal(d2, t2.t())
    for got, g in ((d1, g1), (d2, g2)):
        ref = _ref(x, g)
        assert float((got.double().cpu() - ref).abs().max()) <= 2e-3 * float(ref.abs().max() if m else 0.0) + 1e-3
    # fp32 operands (no MFMA form): the same public call falls back to two products
    a, b = dense.grad_weight_shared_x(x.float(), g1.float(), g2.float())
    if m:
        assert float((a.double().cpu() - _ref(x, g1)).abs().max()) <= 1e-3 * float(_ref(x, g1).abs().max()) + 1e-3
    assert a.shape == (k, n1) and b.shape == (k, n2)


def test_wide_operands_pair_their_column_blocks():
    """grad_weight_pair with 602-column operands (the first layer at the Reddit shape): the 256-column blocks of both operands are
    paired two per launch and written straight into the [K, N] destinations -- same results as one operand after the other."""
    from dgll_amd import dense, ops

    dev = torch.device("cuda:0")
    torch.manual_seed(9)
    m, k, n = 30000, 602, 256
    x1 = ops.alloc_features(m, k, torch.bfloat16, dev); x1.copy_(torch.randn(m, k, device=dev))
    x2 = ops.alloc_features(m, k, torch.bfloat16, dev); x2.copy_(torch.randn(m, k, device=dev))
    g = torch.randn(m, n, device=dev).to(torch.bfloat16)
    o1 = torch.full((k, n), float("nan"), device=dev)
    o2 = torch.full((k, n), float("nan"), device=dev)
    d1, d2 = dense.grad_weight_pair(x1, x2, g, out1=o1, out2=o2)
    assert d1 is o1 and d2 is o2
    for got, x in ((d1, x1), (d2, x2)):
        ref = _ref(x, g)
        assert float((got.double().cpu() - ref).abs().max()) <= 2e-3 * float(ref.abs().max())
        assert torch.equal(got, dense.grad_weight(x, g))              # block by block the same launches' arithmetic
    a, b = dense.grad_weight_pair(x1, x2[:, :100], g)                   # unequal widths, fresh outputs
    assert torch.equal(a, d1) and torch.equal(b, d2[:100])
