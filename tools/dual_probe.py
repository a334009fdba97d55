import torch, sys
sys.path.insert(0, "/root/repo")
from dgll_amd import dense
dev = torch.device("cuda:0")
M = 2449029
g = torch.randn(M, 256, device=dev).to(torch.bfloat16)
ws = (torch.randn(256, 256, device=dev) / 16).to(torch.bfloat16)
wn = (torch.randn(256, 256, device=dev) / 16).to(torch.bfloat16)
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ts.sort(); return ts[len(ts) // 2]
print("dual MFMA %.3f ms | two library mm %.3f ms" % (timed(lambda: dense.transform_bf16_dual(g, ws, wn)), timed(lambda: (torch.mm(g, ws.t()), torch.mm(g, wn.t())))))
