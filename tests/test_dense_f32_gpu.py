"""fp32 dense products of the layers on the hand-written matrix-core kernels (csrc/gemm_f32.hip): dgll_hip_mm_f32
(v_mfma_f32_32x32x2_f32) and dgll_hip_grad_weight_f32 against float64 products -- F.mm / F.matmul of gcnconv.py:30,
sageconv.py:41,72, gatconv.py:31,117 and their autograd gradients; plus the bf16 re-layout / wide-output paths of mm_nt."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,k,n", [(1000, 256, 256), (333, 19, 32), (4097, 100, 47), (65, 7, 300), (1, 5, 3)])
@pytest.mark.parametrize("relu", [False, True])
def test_mm_f32_matches_float64(cuda_device, m, k, n, relu):
    from dgll_amd import dense

    torch.manual_seed(m + n)
    a, wt = torch.randn(m, k), torch.randn(n, k)
    bias, addend = torch.randn(n), torch.randn(m, n)
    want = a.double() @ wt.double().T + addend.double() + bias.double()
    want = torch.relu(want) if relu else want
    got = dense.mm_nt(a.to(cuda_device), wt.to(cuda_device), relu=relu, bias=bias.to(cuda_device), addend=addend.to(cuda_device))
    assert got.dtype == torch.float32
    torch.testing.assert_close(got.cpu().double(), want, rtol=1e-5, atol=1e-4)
    # unaligned view (column slice with an odd offset): the scalar-load path
    big = torch.randn(m, k + 3).to(cuda_device)
    got2 = dense.mm_nt(big[:, 1:k + 1], wt.to(cuda_device))
    torch.testing.assert_close(got2.cpu().double(), big[:, 1:k + 1].cpu().double() @ wt.double().T, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("m,k1,k2,n", [(3000, 256, 256, 256), (1031, 100, 100, 256), (777, 47, 47, 200), (300, 33, 7, 47), (5, 3, 260, 9)])
@pytest.mark.parametrize("gated", [False, True])
def test_mm2_f32_two_products_one_accumulation_and_the_gate(cuda_device, m, k1, k2, n, gated):
    """dgll_hip_mm2_f32: gate(act(a1 . w1t^T + a2 . w2t^T)) against float64 -- sageConv's self + neighbour term (sageconv.py:72-75)
    and the aggregate-first input gradient with the ReLU mask of the layer below in the epilogue; reduction lengths that are not
    multiples of the 32-wide LDS chunk or of 8, unaligned rows (odd leading dimensions), row counts that are not multiples of 256."""
    from dgll_amd import dense

    torch.manual_seed(m + n)
    a1, a2 = torch.randn(m, k1), torch.randn(m, k2)
    w1, w2 = torch.randn(n, k1), torch.randn(n, k2)
    gate = torch.randn(m, n) if gated else None
    want = a1.double() @ w1.double().T + a2.double() @ w2.double().T
    want = torch.relu(want) if not gated else want
    if gated:
        want = want * (gate > 0).double()
    d = cuda_device
    got = dense.mm2_nt(a1.to(d), w1.to(d), a2.to(d), w2.to(d), relu=not gated, gate=gate.to(d) if gated else None)
    assert got.dtype == torch.float32 and got.shape == (m, n)
    torch.testing.assert_close(got.cpu().double(), want, rtol=1e-5, atol=2e-4)
    # the same through column-sliced (unaligned) views of wider buffers
    b1, b2 = torch.randn(m, k1 + 3).to(d), torch.randn(m, k2 + 5).to(d)
    got2 = dense.mm2_nt(b1[:, 1:k1 + 1], w1.to(d), b2[:, 3:k2 + 3], w2.to(d))
    want2 = b1[:, 1:k1 + 1].cpu().double() @ w1.double().T + b2[:, 3:k2 + 3].cpu().double() @ w2.double().T
    torch.testing.assert_close(got2.cpu().double(), want2, rtol=1e-5, atol=2e-4)
    assert torch.equal(got, dense.mm2_nt(a1.to(d), w1.to(d), a2.to(d), w2.to(d), relu=not gated, gate=gate.to(d) if gated else None))


@pytest.mark.parametrize("m,k,n", [(5000, 64, 47), (100000, 100, 256), (17, 300, 5), (1, 3, 2), (70000, 256, 256), (4099, 257, 300), (9, 31, 33)])
def test_grad_weight_f32_matches_float64_and_is_deterministic(cuda_device, m, k, n):
    from dgll_amd import dense

    torch.manual_seed(m)
    x, g = torch.randn(m, k), torch.randn(m, n)
    want = x.double().T @ g.double()
    got = dense.grad_weight(x.to(cuda_device), g.to(cuda_device))
    assert got.dtype == torch.float32 and got.shape == (k, n)
    scale = float(want.abs().max()) + 1e-9
    assert float((got.cpu().double() - want).abs().max()) <= 2e-5 * scale * max(1.0, (m / 1e4) ** 0.5)
    assert torch.equal(got, dense.grad_weight(x.to(cuda_device), g.to(cuda_device)))


def test_fp32_layers_run_without_library_gemms(cuda_device):
    """gcnConv / sageConv in fp32 on the GPU: forward and gradients equal the CPU layer (the 1e-4 parity claim), every dense
    product on the hand-written kernels (F.mm / F.matmul are routed by dgll_amd.backend)."""
    from dgll_amd import nn as dnn, synth

    torch.manual_seed(3)
    g_cpu = synth.rmat_graph(9, 8, seed=1, device="cpu", symmetric=True)
    x = torch.randn(g_cpu.n_rows, 33)
    layer = dnn.gcnConv(33, 16)
    ref = layer(x.clone().requires_grad_(), g_cpu)
    ref.sum().backward()
    want_w = layer.weight.grad.clone()
    layer.zero_grad()
    got = layer.to(cuda_device)(x.to(cuda_device).requires_grad_(), g_cpu.to(cuda_device))
    got.sum().backward()
    torch.testing.assert_close(got.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(layer.weight.grad.cpu(), want_w, rtol=1e-3, atol=1e-3)


def test_bf16_mm_nt_relayout_and_wide_outputs(cuda_device):
    """Rows that are not 16-byte aligned are re-laid out, more than 256 output columns run as 256-column launches."""
    from dgll_amd import dense

    torch.manual_seed(0)
    a = torch.randn(3000, 47).to(torch.bfloat16)                      # 94-byte rows: not 16-byte aligned
    wt = (torch.randn(300, 47) * 0.2).to(torch.bfloat16)
    got = dense.mm_nt(a.to(cuda_device), wt.to(cuda_device))
    want = a.float() @ wt.float().T
    assert got.shape == (3000, 300) and got.dtype == torch.bfloat16
    assert float((got.float().cpu() - want).abs().max()) <= 2e-2 * float(want.abs().max())
    gw = dense.grad_weight(a.to(cuda_device), got)                   # [47, 300]: unaligned x, wide g
    want_gw = a.float().T @ got.float().cpu()
    assert float((gw.cpu() - want_gw).abs().max()) <= 2e-2 * float(want_gw.abs().max())
