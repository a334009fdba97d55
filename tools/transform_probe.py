"""MFMA transform kernel vs the library sequence it replaces, at the bench shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import dense, ops  # noqa: E402

dev = torch.device("cuda:0")
M = 2_449_029


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for K, N in [(256, 256), (100, 256), (256, 47)]:
    h = ops.alloc_features(M, K, torch.bfloat16, dev); h.copy_(torch.randn(M, K, device=dev))
    agg = ops.alloc_features(M, K, torch.bfloat16, dev); agg.copy_(torch.randn(M, K, device=dev))
    ws = torch.randn(K, N, device=dev).to(torch.bfloat16); wn = torch.randn(K, N, device=dev).to(torch.bfloat16)
    wst, wnt = ws.t().contiguous(), wn.t().contiguous()

    def lib():
        out = torch.addmm(torch.mm(h, ws), agg, wn)
        return out.relu_()

    ideal = (2 * M * K + M * N) * 2 / 5e12 * 1e3
    from dgll_amd import _lib
    for variant in (1, 0, 1, 0):   # 1 = the 4-wave kernel, 0 = the shipped choice (the resident-weights kernel)
        _lib.lib.dgll_hip_debug_tune(4, variant)
        tf, ts = t(lambda: dense.transform_bf16(h, wst, agg, wnt, relu=True)), t(lambda: dense.transform_bf16(h, wst))
        print("   variant=%d  fused %.3f ms (%.2f TB/s)  single %.3f ms (%.2f TB/s)" % (
            variant, tf, (2 * M * K + M * N) * 2 / tf / 1e9, ts, (M * K + M * N) * 2 / ts / 1e9), flush=True)
    _lib.lib.dgll_hip_debug_tune(4, 0)
    print("K=%d N=%d  library mm+addmm+relu %.3f ms | MFMA fused %.3f ms | single-pair MFMA %.3f ms vs mm %.3f ms | ideal@5TB/s %.3f" % (
        K, N, t(lib), t(lambda: dense.transform_bf16(h, wst, agg, wnt, relu=True)),
        t(lambda: dense.transform_bf16(h, wst)), t(lambda: torch.mm(h, ws)), ideal), flush=True)
