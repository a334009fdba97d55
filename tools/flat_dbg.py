import sys, torch
sys.path.insert(0, '/root/repo')
import dgll_amd
from dgll_amd import _lib, ops
dev = torch.device("cuda:0")
def tune(k, v): _lib.check(_lib.lib.dgll_hip_debug_tune(k, v), "tune")
n = 1 << 20
x = ops.alloc_features(n, 256, torch.bfloat16, dev); x.copy_(torch.randn(n, 256, device=dev).to(torch.bfloat16))
def run(g, label):
    out = {}
    for mode in (1, 0):
        tune(13, mode)
        for _ in range(2): ops.spmm_raw(g, x, reduce="mean")
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3): ops.spmm_raw(g, x, reduce="mean")
        b.record(); torch.cuda.synchronize()
        out[mode] = a.elapsed_time(b) / 3
    print("%-50s old %.3f ms  flat %.3f ms" % (label, out[1], out[0]))
for E in (256, 1024):
    tune(14, E)
    for deg in (8, 50, 64, 200):
        rows = (20_000_000 // deg)
        rp = torch.arange(0, rows + 1, device=dev, dtype=torch.int64) * deg
        col = torch.randint(0, n, (rows * deg,), device=dev, dtype=torch.int32)
        g = dgll_amd.CSRGraph(rp, col, None, rows, n, check=False)
        run(g, "E=%d uniform degree %d, %d rows" % (E, deg, rows))
    # a few long rows among short ones
    deg = torch.full((400_000,), 40, device=dev, dtype=torch.int64)
    deg[::1000] = 5000
    rp = torch.zeros(deg.numel() + 1, device=dev, dtype=torch.int64); torch.cumsum(deg, 0, out=rp[1:])
    col = torch.randint(0, n, (int(rp[-1]),), device=dev, dtype=torch.int32)
    run(dgll_amd.CSRGraph(rp, col, None, deg.numel(), n, check=False), "E=%d degree 40 + every 1000th row 5000" % E)
    deg = torch.full((400_000,), 40, device=dev, dtype=torch.int64)
    deg[::2] = 0
    rp = torch.zeros(deg.numel() + 1, device=dev, dtype=torch.int64); torch.cumsum(deg, 0, out=rp[1:])
    col = torch.randint(0, n, (int(rp[-1]),), device=dev, dtype=torch.int32)
    run(dgll_amd.CSRGraph(rp, col, None, deg.numel(), n, check=False), "E=%d degree 40, every other row empty" % E)
