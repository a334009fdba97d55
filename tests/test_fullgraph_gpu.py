"""Full-graph GraphSAGE (the bench path) on the GPU vs the oracle, and the partitioned engine at world_size 1."""
import numpy as np
import pytest
import torch

from oracle import cref

pytestmark = pytest.mark.gpu


def _setup(dev, n_scale=10, fin=20, hidden=(32, 16)):
    from dgll_amd import nn as dnn
    from dgll_amd import synth

    torch.manual_seed(0)
    g_cpu = synth.rmat_graph(n_scale, 8, seed=2, device="cpu", symmetric=True, weighted=False, self_loops=False)
    model = dnn.GraphSage(fin, list(hidden), None)
    x = torch.randn(g_cpu.n_rows, fin)
    return g_cpu, model, x


def _oracle_forward(g_cpu, model, x):
    h = x.numpy()
    rp, col = g_cpu.rowptr.numpy(), g_cpu.col.numpy()
    for layer in model.gcn:
        agg = cref.spmm_csr(rp, col, None, h, reduce="mean")
        h = np.maximum(cref.gemm(h, layer.weight.detach().numpy()) + cref.gemm(agg, layer.neighborAgg.weight.detach().numpy()), 0)
    return h


def test_forward_graph_matches_oracle_and_autograd(cuda_device):
    g_cpu, model, x = _setup(cuda_device)
    ref = _oracle_forward(g_cpu, model, x)
    # CPU autograd of the same formula for the gradients
    xr = x.clone().requires_grad_()
    row = g_cpu.row_index()
    deg = g_cpu.degrees().clamp(min=1).float()
    h = xr
    for layer in model.gcn:
        agg = torch.zeros(g_cpu.n_rows, h.shape[1]).index_add_(0, row, h[g_cpu.col.long()]) / deg[:, None]
        h = torch.relu(h @ layer.weight + agg @ layer.neighborAgg.weight)
    gout = torch.randn_like(h)
    (h * gout).sum().backward()
    ref_grads = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()

    m = model.to(cuda_device)
    xd = x.to(cuda_device).requires_grad_()
    out = m.forward_graph(g_cpu.to(cuda_device), xd)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
    (out * gout.to(cuda_device)).sum().backward()
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=2e-3, atol=2e-4)
    for p, r in zip(m.parameters(), ref_grads):
        np.testing.assert_allclose(p.grad.cpu().numpy(), r.numpy(), rtol=2e-3, atol=5e-4)


def test_world1_partition_engine_equals_forward_graph(cuda_device):
    from dgll_amd import dist as ddist

    g_cpu, model, x = _setup(cuda_device)
    g = g_cpu.to(cuda_device)
    m = model.to(cuda_device)
    part = ddist.partition_contiguous(g, 1, 0)
    assert part.n_halo == 0 and part.local.nnz == g.nnz
    engine = ddist.DistGraph(part, cuda_device)
    xd = x.to(cuda_device)
    a = m.forward_graph(g, xd)
    b = engine.sage_forward(m, engine.permute_to_local(xd))
    assert torch.equal(a, b)
    # RaCoM at world 1 is the identity on the gradients
    b.sum().backward()
    before = [p.grad.clone() for p in m.parameters()]
    ddist.RaCoM(m.parameters(), cuda_device).all_reduce_and_wait()
    for p, q in zip(m.parameters(), before):
        assert torch.equal(p.grad, q)


def test_bf16_forward_graph_tracks_fp32(cuda_device):
    from dgll_amd import ops

    g_cpu, model, x = _setup(cuda_device, fin=100, hidden=(256, 47))
    g = g_cpu.to(cuda_device)
    m = model.to(cuda_device)
    xd = x.to(cuda_device)
    ref = m.forward_graph(g, xd)
    xb = ops.alloc_features(g.n_rows, 100, torch.bfloat16, cuda_device)
    xb.copy_(xd)
    out = m.forward_graph(g, xb)
    assert out.dtype == torch.bfloat16
    err = (out.float() - ref).abs().max() / ref.abs().max()
    assert float(err) < 3e-2, float(err)
    out.float().sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())


def test_bf16_fused_relu_gates_equal_the_unfused_layers(cuda_device):
    """The bench configuration (100-256-256-47, bf16): forward_graph hands gradients between its layers already masked
    (SpMM gate epilogue, MFMA output gate) and skips the separate ReLU-backward passes.  Parameter and input gradients
    must equal those of the same layers run one by one without that agreement."""
    from dgll_amd import fused_layers, ops

    g_cpu, model, x = _setup(cuda_device, fin=100, hidden=(256, 256, 47))
    g = g_cpu.to(cuda_device)
    m = model.to(cuda_device)
    xb = ops.alloc_features(g.n_rows, 100, torch.bfloat16, cuda_device)
    xb.copy_(x.to(cuda_device))
    gout = torch.randn(g.n_rows, 47, device=cuda_device).to(torch.bfloat16)

    def run(fused):
        m.zero_grad()
        store = ops.alloc_features(g.n_rows, 100, torch.bfloat16, cuda_device)
        store.copy_(xb)
        xin = store.requires_grad_()
        if fused:
            out = m.forward_graph(g, xin)
        else:
            h = xin
            for layer in m.gcn:
                h = fused_layers.sage_graph_layer(layer, g, h)          # no gate agreement: every node masks for itself
            out = h
        (out.float() * gout.float()).sum().backward()
        return out.detach(), xin.grad.detach().clone(), [p.grad.detach().clone() for p in m.parameters()]

    out_a, gx_a, gp_a = run(True)
    out_b, gx_b, gp_b = run(False)
    assert torch.equal(out_a, out_b)
    # same kernels and operand values; the only difference is where the mask is applied -> agreement to bf16 rounding
    torch.testing.assert_close(gx_a.float(), gx_b.float(), rtol=2e-2, atol=2e-2 * float(gx_b.float().abs().max()))
    for a, b in zip(gp_a, gp_b):
        torch.testing.assert_close(a, b, rtol=2e-2, atol=2e-2 * float(b.abs().max()))


def test_mfma_transform_output_gate(cuda_device):
    from dgll_amd import dense, ops

    M, K1, K2, N = 1111, 47, 47, 256
    a1 = ops.alloc_features(M, K1, torch.bfloat16, cuda_device, pad_to=64)
    a1.copy_(torch.randn(M, K1, device=cuda_device))
    a2 = ops.alloc_features(M, K2, torch.bfloat16, cuda_device)
    a2.copy_(torch.randn(M, K2, device=cuda_device))
    w1 = torch.randn(N, K1, device=cuda_device).to(torch.bfloat16)
    w2 = torch.randn(N, K2, device=cuda_device).to(torch.bfloat16)
    gate = torch.randn(M, N, device=cuda_device).to(torch.bfloat16)
    gate[5] = 0
    ref = (a1.float() @ w1.float().t() + a2.float() @ w2.float().t()) * (gate.float() > 0)
    out = dense.transform_bf16(a1, w1, a2, w2, out_dtype=torch.float32, out_gate=gate)
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-3)
    rs = torch.rand(M, device=cuda_device) + 0.5
    out_rs = dense.transform_bf16(a1, w1, a2, w2, out_dtype=torch.float32, out_gate=gate, row_scale=rs)
    np.testing.assert_allclose(out_rs.cpu().numpy(), (ref * rs[:, None]).cpu().numpy(), rtol=1e-4, atol=1e-3)
    out_rs16 = dense.transform_bf16(a1, w1, row_scale=rs, relu=True)
    np.testing.assert_allclose(out_rs16.float().cpu().numpy(), ((a1.float() @ w1.float().t()) * rs[:, None]).relu().cpu().numpy(),
                               rtol=1e-2, atol=5e-2)
    out16 = dense.transform_bf16(a1, w1, a2, w2, out_gate=gate)
    assert bool(((out16 == 0) | (gate > 0)).all())
    np.testing.assert_allclose(out16.float().cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol=5e-2)


def test_split_k_weight_gradient(cuda_device):
    from dgll_amd import dense

    x = torch.randn(70_001, 24, device=cuda_device)
    g = torch.randn(70_001, 40, device=cuda_device)
    np.testing.assert_allclose(dense.grad_weight(x, g).cpu().numpy(), (x.double().t() @ g.double()).cpu().numpy(), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("M,K1,K2,N,relu", [(1000, 256, 256, 256, True), (777, 100, 100, 256, True), (513, 256, 256, 47, False),
                                            (130, 64, 0, 128, False), (4097, 40, 24, 33, True), (128, 256, 0, 256, False),
                                            (3000, 64, 0, 10, False), (500, 40, 40, 64, True), (77, 8, 0, 1, False)])
def test_mfma_transform_kernel(cuda_device, M, K1, K2, N, relu):
    """dgll_hip_transform_bf16 (v_mfma_f32_32x32x16_bf16) vs an fp32 matmul of the same bf16-rounded operands computed ON
    THE HOST (torch CPU: shares neither the device nor a BLAS with the kernel under test); the weights are asymmetric so a
    transposed fragment layout cannot pass."""
    from dgll_amd import dense, ops

    torch.manual_seed(M + N)
    a1 = ops.alloc_features(M, K1, torch.bfloat16, cuda_device)
    a1.copy_(torch.randn(M, K1, device=cuda_device))
    w1 = (torch.randn(K1, N, device=cuda_device) * 0.1 + torch.arange(N, device=cuda_device) * 1e-3).to(torch.bfloat16)
    ref = a1.float().cpu() @ w1.float().cpu()
    a2 = w2 = None
    if K2:
        a2 = ops.alloc_features(M, K2, torch.bfloat16, cuda_device)
        a2.copy_(torch.randn(M, K2, device=cuda_device))
        w2 = (torch.randn(K2, N, device=cuda_device) * 0.1).to(torch.bfloat16)
        ref = ref + a2.float().cpu() @ w2.float().cpu()
    if relu:
        ref = ref.relu()
    out32 = dense.transform_bf16(a1, w1.t(), a2, None if w2 is None else w2.t(), relu=relu, out_dtype=torch.float32)
    np.testing.assert_allclose(out32.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-3)
    out = dense.transform_bf16(a1, w1.t(), a2, None if w2 is None else w2.t(), relu=relu)
    assert out.dtype == torch.bfloat16 and out.shape == (M, N)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol=2e-2)


def test_mfma_transform_relu_mask_and_nan_pads(cuda_device):
    """Fused ReLU-backward mask; uninitialised pad columns (NaN bit patterns) must not leak into the result."""
    from dgll_amd import dense

    M, K, N = 300, 100, 64
    store = torch.full((M, 104), float("nan"), device=cuda_device, dtype=torch.bfloat16)
    a = store[:, :K]
    a.copy_(torch.randn(M, K, device=cuda_device))
    mstore = torch.full((M, 104), float("nan"), device=cuda_device, dtype=torch.bfloat16)
    mask = mstore[:, :K]
    mask.copy_(torch.randn(M, K, device=cuda_device))
    w = torch.randn(K, N, device=cuda_device).to(torch.bfloat16)
    out = dense.transform_bf16(a, w.t(), out_dtype=torch.float32, mask=mask)
    ref = (a.float() * (mask.float() > 0)) @ w.float()
    assert torch.isfinite(out).all()
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("K,N,ld_align,extra", [(256, 47, 64, ""), (256, 47, 64, "addend"), (256, 47, 8, "gate"), (640, 47, 64, ""),
                                                (100, 33, 64, "addend"), (64, 8, 64, ""), (256, 130, 8, "")])
def test_mfma_transform_writes_the_row_padding_with_zeros(cuda_device, K, N, ld_align, extra):
    """relu bit 1 of dgll_hip_transform_bf16* (set by dense.transform_bf16, which allocates the output): rows are stored as whole
    16-byte vectors up to the padded width; the padding columns come out as exact zeros whatever the addend's / gate's own padding
    holds, the [M, N] result is unchanged.  Without the bit (a caller's own buffer) the padding is not touched."""
    from dgll_amd import _lib, dense

    dev = cuda_device
    torch.manual_seed(K + N)
    M = 1000
    a = dense._as_rows16(torch.randn(M, K, device=dev).to(torch.bfloat16))      # 16-byte aligned rows (K = 100: 104-column pitch)
    w = (torch.randn(N, K, device=dev) * 0.1).to(torch.bfloat16)
    ld = -(-N // ld_align) * ld_align
    side = torch.full((M, ld), float("nan"), device=dev, dtype=torch.bfloat16)      # addend / gate with NaN in ITS padding
    side[:, :N] = torch.randn(M, N, device=dev)
    kw = {"addend": side[:, :N]} if extra == "addend" else {"out_gate": side[:, :N]} if extra == "gate" else {}
    out = dense.transform_bf16(a, w, relu=True, ld_align=ld_align, **kw)
    z = a.float() @ w.float().t()
    if extra == "addend":
        z = z + side[:, :N].float()
    ref = torch.relu(z)
    if extra == "gate":
        ref = ref * (side[:, :N].float() > 0)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.cpu().numpy(), rtol=1e-2, atol=2e-2)
    whole = out.as_strided((M, out.stride(0)), (out.stride(0), 1))
    assert out.stride(0) == ld and bool((whole[:, N:] == 0).all())
    # the C entry point without the bit, into a caller's buffer: nothing beyond column N is written
    mine = torch.full((M, ld), float("nan"), device=dev, dtype=torch.bfloat16)
    p1 = dense._pad_wt(w)
    code = _lib.lib.dgll_hip_transform_bf16(torch.cuda.current_stream(dev).cuda_stream, a.data_ptr(), a.stride(0), K, p1.data_ptr(),
                                            p1.stride(0), None, 0, 0, None, 0, p1.shape[0], None, 0, mine.data_ptr(), ld, _lib.BF16, M, N, 1,
                                            None)
    _lib.check(code, "dgll_hip_transform_bf16")
    assert bool(torch.isnan(mine[:, N:]).all()) and bool(torch.isfinite(mine[:, :N]).all())
    if not extra:
        assert torch.equal(mine[:, :N], out)


def test_bf16_bench_configuration_against_the_storage_emulating_oracle(cuda_device):
    """The bench configuration (100-256-256-47, bf16, full graph) against CPU autograd of oracle/torch_ref.sage_block with bf16
    rounding applied exactly where the GPU path stores a tensor: every layer's stored output must agree entry for entry (up to
    fp32 accumulation order at a rounding boundary) and the gradients of forward_graph -- gates fused into the SpMM / MFMA
    epilogues -- differ by the bf16 rounding of stored gradients only.  (The same construction at the Reddit shape exposed a
    two-launch fallback that rounded one product before the add: tests/test_config2_reddit_gpu.py.)"""
    from dgll_amd import fused_layers, ops
    from oracle import torch_ref

    dev = cuda_device
    g_cpu, model, x = _setup(dev, n_scale=13, fin=100, hidden=(256, 256, 47))
    g = g_cpu.to(dev)
    m = model.to(dev)
    rnd = lambda t: t.to(torch.bfloat16).float()                     # noqa: E731
    store = lambda t: t + (rnd(t.detach()) - t.detach())             # noqa: E731  (straight-through gradient)
    xb = ops.alloc_features(g.n_rows, 100, torch.bfloat16, dev)
    xb.copy_(x.to(dev))
    gout = torch.randn(g.n_rows, 47).to(torch.bfloat16)

    # ---- oracle
    cx = xb.float().cpu().requires_grad_()
    params, hid, outs = [], cx, []
    for layer in m.gcn:
        ws = rnd(layer.weight.detach().cpu()).requires_grad_()
        wn = rnd(layer.neighborAgg.weight.detach().cpu()).requires_grad_()
        params.append((ws, wn))
        hid = torch_ref.sage_block(g_cpu.rowptr, g_cpu.col, hid, hid, ws, wn, act=layer.activation is not None,
                                   transform_first=layer.hidden_dim < layer.input_dim, store=store)
        outs.append(hid)
    (hid * gout.float()).sum().backward()

    # ---- layer by layer: stored outputs
    with torch.no_grad():
        h = xb
        for l, layer in enumerate(m.gcn):
            h = fused_layers.sage_graph_layer(layer, g, h)
            ref = outs[l].detach()
            equal = float((h.float().cpu() == ref).float().mean())
            flips = float(((h.float().cpu() > 0) != (ref > 0)).float().mean()) if layer.activation is not None else 0.0
            print("layer %d: %.5f of the stored outputs identical, gates that differ %.1e" % (l, equal, flips))
            assert equal >= 0.999 and flips <= 2e-5, (l, equal, flips)            # measured 0.99955 ... 0.99997, no gate

    # ---- the fused step: gradients
    store_x = ops.alloc_features(g.n_rows, 100, torch.bfloat16, dev)
    store_x.copy_(xb)
    xin = store_x.requires_grad_()
    out = m.forward_graph(g, xin)
    assert float((out.float().cpu() == outs[-1].detach()).float().mean()) >= 0.995
    (out.float() * gout.to(dev).float()).sum().backward()

    def close(name, a, ref, tol=1e-2):                               # measured 0.7e-4 ... 3.7e-3
        a, ref = a.detach().float().cpu(), ref.detach()
        rel = float((a - ref).norm() / ref.norm())
        print("%-28s relative L2 error %.3e" % (name, rel))
        assert rel <= tol, (name, rel)

    for l, layer in enumerate(m.gcn):
        close("layer %d weight" % l, layer.weight.grad, params[l][0].grad)
        close("layer %d neighborAgg.weight" % l, layer.neighborAgg.weight.grad, params[l][1].grad)
    close("input features", xin.grad, cx.grad)


def test_bf16_gcn_against_the_storage_emulating_oracle(cuda_device):
    """Two-layer GCN (gcnconv.py:43-58) in bf16 on a row-normalised weighted adjacency against CPU autograd with bf16 rounding where
    the GPU path stores a tensor (x.W, the aggregated + biased + activated rows): stored hidden rows identical almost everywhere,
    log-probabilities (fp32) and every gradient within bf16 rounding of the oracle."""
    from dgll_amd import nn as dnn
    from dgll_amd import ops, synth

    dev = cuda_device
    torch.manual_seed(0)
    g_cpu = synth.rmat_graph(12, 10, seed=3, device="cpu", symmetric=True, weighted=True, self_loops=True)
    n, fin, nhid, ncls = g_cpu.n_rows, 100, 256, 47
    model = dnn.GCN(fin, nhid, ncls, dropout=0.0).to(dev)
    xb = ops.alloc_features(n, fin, torch.bfloat16, dev)
    xb.copy_(torch.randn(n, fin, device=dev))
    gout = torch.randn(n, ncls)
    rnd = lambda t: t.to(torch.bfloat16).float()                     # noqa: E731

    class _Store(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return rnd(t)

        @staticmethod
        def backward(ctx, gr):
            return rnd(gr)

    store = _Store.apply
    adj = torch.sparse_csr_tensor(g_cpu.rowptr, g_cpu.col.long(), g_cpu.val.float(), (n, n))
    cx = xb.float().cpu().requires_grad_()
    W1, b1 = rnd(model.gcn1.weight.detach().cpu()).requires_grad_(), model.gcn1.bias.detach().cpu().clone().requires_grad_()
    W2, b2 = rnd(model.gcn2.weight.detach().cpu()).requires_grad_(), model.gcn2.bias.detach().cpu().clone().requires_grad_()
    h1 = store(torch.relu(torch.sparse.mm(adj, store(cx @ W1)) + b1))
    ref = torch.log_softmax(store(torch.sparse.mm(adj, store(h1 @ W2)) + b2), dim=1)
    (ref * gout).sum().backward()

    xin = ops.alloc_features(n, fin, torch.bfloat16, dev)
    xin.copy_(xb)
    xin.requires_grad_()
    graph = g_cpu.to(dev)
    with torch.no_grad():
        hid = model.gcn1(xin, graph, relu=True).float().cpu()
    equal = float((hid == h1.detach()).float().mean())
    print("hidden rows: %.5f identical" % equal)
    assert equal >= 0.999
    out = model(xin, graph)
    assert out.dtype == torch.float32
    err = (out.detach().cpu() - ref.detach()).abs()
    assert float((err <= 1e-4 * (1.0 + ref.detach().abs())).float().mean()) >= 0.98 and float(err.max()) <= 0.1
    (out * gout.to(dev)).sum().backward()

    def close(name, a, r, tol=1e-2):
        rel = float((a.detach().float().cpu() - r).norm() / r.norm())
        print("%-12s relative L2 error %.3e" % (name, rel))
        assert rel <= tol, (name, rel)

    close("gcn1.weight", model.gcn1.weight.grad, W1.grad, 1e-3)       # measured 1e-5 ... 3e-5: the oracle rounds the stored
    close("gcn1.bias", model.gcn1.bias.grad, b1.grad, 1e-3)           # gradients where the GPU path does
    close("gcn2.weight", model.gcn2.weight.grad, W2.grad, 1e-3)
    close("gcn2.bias", model.gcn2.bias.grad, b2.grad, 1e-3)
    close("input", xin.grad, cx.grad, 1e-2)                           # measured 1.7e-3 (its own bf16 rounding)


@pytest.mark.parametrize("n,k,dtype,transposed", [(47, 256, torch.float32, True), (256, 100, torch.float32, True), (130, 602, torch.bfloat16, True),
                                                  (8, 48, torch.float32, False), (256, 256, torch.bfloat16, False), (1, 1, torch.float32, True)])
def test_weight_packing_is_cast_pad_and_layout_in_one_launch(cuda_device, n, k, dtype, transposed):
    """dense._pad_wt (dgll_hip_pack_weight_bf16): the zero-padded bf16 block the transform kernels stage, from a parameter or its
    transposed view, fp32 or bf16 -- equal to casting, zero-filling and copying with torch."""
    from dgll_amd import dense

    torch.manual_seed(n * 1000 + k)
    base = torch.randn((k, n) if transposed else (n, k), device=cuda_device).to(dtype)
    wt = base.t() if transposed else base
    got = dense._pad_wt(wt)
    rows = 64 if n <= 64 else 128 if n <= 128 else 256
    want = torch.zeros((rows, -(-k // 64) * 64), dtype=torch.bfloat16, device=cuda_device)
    want[:n, :k] = wt.to(torch.bfloat16)
    assert got.shape == want.shape and got.dtype == torch.bfloat16 and torch.equal(got, want)
    assert torch.equal(dense._pad_wt(wt, rows=256)[:rows], want[:rows]) and bool((dense._pad_wt(wt, rows=256)[n:] == 0).all())
    # wcast: the fp32 parameter itself on the bf16 GPU path, a cast otherwise
    act = torch.empty(1, dtype=torch.bfloat16, device=cuda_device)
    assert dense.wcast(base, act) is base                      # fp32: packed (and cast) later; bf16: nothing to do
    assert dense.wcast(base, act.float()).dtype == torch.float32


def _sign_bits(t):
    """Reference packing of (t > 0): int32 [M, bit_words(N)], bit b of word w = column 32 w + b."""
    from dgll_amd import dense

    m, n = t.shape
    words = dense.bit_words(n)
    pos = torch.zeros(m, words * 32, dtype=torch.int64, device=t.device)
    pos[:, :n] = (t.float() > 0).to(torch.int64)
    w = (pos.view(m, words, 32) << torch.arange(32, device=t.device)).sum(-1)           # < 2^32
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)


@pytest.mark.parametrize("M,K1,K2,N", [(1000, 256, 256, 256), (777, 100, 100, 256), (513, 256, 0, 47), (130, 64, 0, 128),
                                       (4097, 40, 24, 33), (2049, 256, 256, 200), (77, 8, 0, 1), (600, 320, 320, 256)])
def test_mfma_transform_sign_bits_of_the_output(cuda_device, M, K1, K2, N):
    """dgll_hip_transform_bf16_bits, producer side: the bits are those of the bf16 values the same launch stored, the values
    themselves are bit-equal to the launch without bits; padding words / bits past N are zero.  (600, 320+320) runs the 4-wave
    kernel: the bits come from its one-pass fallback there."""
    from dgll_amd import dense, ops

    a1 = ops.alloc_features(M, K1, torch.bfloat16, cuda_device)
    a1.copy_(torch.randn(M, K1, device=cuda_device))
    a2 = w2 = None
    if K2:
        a2 = ops.alloc_features(M, K2, torch.bfloat16, cuda_device)
        a2.copy_(torch.randn(M, K2, device=cuda_device))
        w2 = torch.randn(N, K2, device=cuda_device).to(torch.bfloat16)
    w1 = torch.randn(N, K1, device=cuda_device).to(torch.bfloat16)
    for relu in (True, False):
        want = dense.transform_bf16(a1, w1, a2, w2, relu=relu)
        out, bits = dense.transform_bf16(a1, w1, a2, w2, relu=relu, bits_out=True)
        assert torch.equal(out, want)
        assert bits.dtype == torch.int32 and bits.shape == (M, dense.bit_words(N))
        assert torch.equal(bits, _sign_bits(out))


@pytest.mark.parametrize("M,K1,K2,N", [(1111, 47, 47, 256), (1000, 256, 256, 256), (513, 64, 0, 128), (4097, 40, 24, 33),
                                       (300, 128, 0, 64), (2049, 256, 256, 200), (600, 320, 320, 256)])
def test_mfma_transform_gate_given_as_bits(cuda_device, M, K1, K2, N):
    """Consumer side: the gate as bits gives bit-for-bit what the gate as a bf16 matrix gives -- with and without out_gate passed
    along, into an own allocation (padding written) and into a caller's buffer."""
    from dgll_amd import dense, ops

    a1 = ops.alloc_features(M, K1, torch.bfloat16, cuda_device, pad_to=64)
    a1.copy_(torch.randn(M, K1, device=cuda_device))
    a2 = w2 = None
    if K2:
        a2 = ops.alloc_features(M, K2, torch.bfloat16, cuda_device)
        a2.copy_(torch.randn(M, K2, device=cuda_device))
        w2 = torch.randn(N, K2, device=cuda_device).to(torch.bfloat16)
    w1 = torch.randn(N, K1, device=cuda_device).to(torch.bfloat16)
    gate = ops.alloc_features(M, N, torch.bfloat16, cuda_device)                      # rows on a 16-byte pitch, as every activation
    gate.copy_(torch.relu(torch.randn(M, N, device=cuda_device)))                     # what a ReLU leaves: zeros and positives
    gate[5] = 0
    gate[7] = 1
    bits = _sign_bits(gate)
    want = dense.transform_bf16(a1, w1, a2, w2, out_gate=gate)
    got = dense.transform_bf16(a1, w1, a2, w2, out_gate=gate, gate_bits=bits)
    assert torch.equal(got, want)
    assert bool((got[5] == 0).all()) and bool(((got == 0) | (gate > 0)).all())
    if K1 + K2 <= 512:                                         # the resident-weights kernel: the bits alone are enough
        assert torch.equal(dense.transform_bf16(a1, w1, a2, w2, gate_bits=bits), want)
    else:
        with pytest.raises(RuntimeError, match="out_gate"):
            dense.transform_bf16(a1, w1, a2, w2, gate_bits=bits)
    buf = torch.full((M, N + 8), 7.0, dtype=torch.bfloat16, device=cuda_device)
    if (N + 8) % 8 == 0:
        dense.transform_bf16(a1, w1, a2, w2, out_gate=gate, gate_bits=bits, out=buf[:, :N])
        assert torch.equal(buf[:, :N], want) and bool((buf[:, N:] == 7).all())
    # both at once: gated output and its own sign bits
    out2, bits2 = dense.transform_bf16(a1, w1, a2, w2, out_gate=gate, gate_bits=bits, bits_out=True)
    assert torch.equal(out2, want) and torch.equal(bits2, _sign_bits(want))


def test_gate_bits_argument_checks(cuda_device):
    from dgll_amd import dense, ops

    a = ops.alloc_features(64, 64, torch.bfloat16, cuda_device)
    a.zero_()
    w = torch.zeros(64, 64, device=cuda_device, dtype=torch.bfloat16)
    with pytest.raises(ValueError, match="bits_out"):
        dense.transform_bf16(a, w, out_dtype=torch.float32, bits_out=True)
    with pytest.raises(ValueError, match="gate_bits"):
        dense.transform_bf16(a, w, gate_bits=torch.zeros(64, 3, dtype=torch.int32, device=cuda_device))
    with pytest.raises(ValueError, match="without out_gate"):
        dense.transform_bf16(a, w, gate_bits=torch.zeros(64, 4, dtype=torch.int32, device=cuda_device),
                             row_scale=torch.ones(64, device=cuda_device))


def test_forward_graph_hands_the_relu_gates_on_as_bits(cuda_device, monkeypatch):
    """forward_graph with the sign bits riding along its activations (the default) against DGLL_GATE_BITS=0 (gates read back as
    bf16 activations): every output, input gradient and parameter gradient bit-equal -- the two forms of the gate are the same
    mask.  An activation edited in place between the layers loses its bits (version check) and the bf16 form is read."""
    from dgll_amd import fused_layers, ops

    g_cpu, model, x = _setup(cuda_device, fin=100, hidden=(256, 256, 47))
    g = g_cpu.to(cuda_device)
    m = model.to(cuda_device)
    xb = ops.alloc_features(g.n_rows, 100, torch.bfloat16, cuda_device)
    xb.copy_(x.to(cuda_device))
    gout = torch.randn(g.n_rows, 47, device=cuda_device).to(torch.bfloat16)
    seen = []
    real = fused_layers.dense.transform_bf16

    def spy(*a, **kw):
        seen.append(kw.get("gate_bits") is not None)
        return real(*a, **kw)

    def run(bits):
        monkeypatch.setattr(fused_layers, "GATE_BITS", bits)
        m.zero_grad()
        store = ops.alloc_features(g.n_rows, 100, torch.bfloat16, cuda_device)
        store.copy_(xb)
        xin = store.requires_grad_()
        out = m.forward_graph(g, xin)
        (out.float() * gout.float()).sum().backward()
        return out.detach(), xin.grad.detach().clone(), [p.grad.detach().clone() for p in m.parameters()]

    monkeypatch.setattr(fused_layers.dense, "transform_bf16", spy)
    a = run(True)
    assert any(seen)                                            # the gated input-gradient launches did take the bits
    seen.clear()
    b = run(False)
    assert not any(seen)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for p, q in zip(a[2], b[2]):
        assert torch.equal(p, q)
    # a stale tag is not used
    h = torch.relu(torch.randn(64, 256, device=cuda_device)).to(torch.bfloat16)
    fused_layers._tag_bits(h, _sign_bits(h))
    monkeypatch.setattr(fused_layers, "GATE_BITS", True)
    assert fused_layers._bits_of(h) is not None
    h.mul_(2)
    assert fused_layers._bits_of(h) is None


@pytest.mark.parametrize("M,K,n", [(1000, 256, 47), (777, 100, 64), (513, 256, 100), (130, 64, 128), (77, 8, 1)])
def test_transform_cat_equals_two_transforms(cuda_device, M, K, n):
    """dense.transform_bf16_cat: both products of one activation matrix from one pass, each starting on a 128-byte line of a shared
    row -- bit-equal to the two single transforms while both run the kernel shape with 512-row blocks (n <= 64); the 256-column
    shape walks the reduction of a row from another chunk (256-row blocks), so there the results agree to bf16 rounding."""
    from dgll_amd import dense, ops

    a = ops.alloc_features(M, K, torch.bfloat16, cuda_device)
    a.copy_(torch.randn(M, K, device=cuda_device))
    w1, w2 = torch.randn(n, K, device=cuda_device), torch.randn(n, K, device=cuda_device)
    first, second = dense.transform_bf16_cat(a, w1, w2.t().contiguous().t())           # (any strides)
    assert first.shape == second.shape == (M, n) and first.stride(0) == second.stride(0) == (128 if n <= 64 else 256)
    assert second.data_ptr() - first.data_ptr() == (128 if n <= 64 else 256)
    want1, want2 = dense.transform_bf16(a, w1), dense.transform_bf16(a, w2)
    if n <= 64:
        assert torch.equal(first, want1) and torch.equal(second, want2)
    else:
        torch.testing.assert_close(first.float(), want1.float(), rtol=1e-2, atol=1e-2 * float(want1.float().abs().max()))
        torch.testing.assert_close(second.float(), want2.float(), rtol=1e-2, atol=1e-2 * float(want2.float().abs().max()))
