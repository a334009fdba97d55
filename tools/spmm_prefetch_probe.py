import os, sys
sys.path.insert(0, "/root/repo")
import torch
from dgll_amd import _lib, ops, synth
dev = torch.device("cuda:0")
g = synth.products_like_graph(dev, seed=0, locality=0.9, exact=True, permute_ids=True).reorder(seed=0)[0]
gt, _ = g.transpose(); g.plan(); gt.plan()
n = g.n_rows
h = torch.randn(n, 256, device=dev).to(torch.bfloat16)
w = torch.rand(g.nnz, device=dev)
y = torch.zeros(n, 256, device=dev, dtype=torch.bfloat16)
def timeit(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for flag in (0, 4, 0, 4):
    _lib.lib.dgll_hip_debug_tune(2, flag)
    t1 = timeit(lambda: ops.spmm_raw(g, h, reduce="mean"))
    t2 = timeit(lambda: ops.spmm_raw(gt, h, val=w, reduce="sum", out=y, accumulate=True, gate=h))
    print("flags %d: mean fwd %.3f ms | weighted accumulate+gate %.3f ms" % (flag, t1, t2), flush=True)
