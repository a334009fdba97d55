"""Probe hipBLASLt layouts for the tall-skinny transforms of the SAGE step (which call is fast?)."""
import torch

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


for K, N in [(256, 256), (100, 256), (256, 47), (104, 256)]:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    wt = w.t().contiguous()
    g = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    print("K=%d N=%d" % (K, N))
    print("  x@w (w [K,N])            %.3f ms" % t(lambda: torch.mm(x, w)))
    print("  linear(x, wt [N,K])      %.3f ms" % t(lambda: torch.nn.functional.linear(x, wt)))
    print("  x@wt.t()                 %.3f ms" % t(lambda: torch.mm(x, wt.t())))
    print("  g@w.t()  (dX)            %.3f ms" % t(lambda: torch.mm(g, w.t())))
    print("  g@wt (dX, wt [N,K])      %.3f ms" % t(lambda: torch.mm(g, wt)))
    print("  x.t()@g  (dW)            %.3f ms" % t(lambda: torch.mm(x.t(), g)))
    flops = 2 * M * K * N
    print("  ideal @5TB/s: %.3f ms ; flops %.1f G" % ((M * K + M * N) * 2 / 5e12 * 1e3, flops / 1e9))
lab = torch.randint(0, 47, (M,), device=dev)
o = torch.randn(M, 47, device=dev, requires_grad=True)
print("cross_entropy sum: %.3f ms" % t(lambda: torch.nn.functional.cross_entropy(o, lab, reduction="sum")))
print("cross_entropy mean: %.3f ms" % t(lambda: torch.nn.functional.cross_entropy(o, lab, reduction="mean")))
print("logsoftmax+gather: %.3f ms" % t(lambda: -torch.log_softmax(o, 1).gather(1, lab[:, None]).sum()))
