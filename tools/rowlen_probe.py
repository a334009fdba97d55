import sys, torch
sys.path.insert(0, '/root/repo')
import dgll_amd
from dgll_amd import ops, synth
dev = torch.device("cuda:0")
g = synth.products_like_graph(dev, seed=0, locality=0.9, exact=True, permute_ids=True)
g, _ = g.reorder(seed=0)
n, nnz = g.n_rows, g.nnz
x = ops.alloc_features(n, 256, torch.bfloat16, dev); x.copy_(torch.randn(n, 256, device=dev).to(torch.bfloat16))
def t(gr, label):
    gr.plan()
    for _ in range(2): ops.spmm_raw(gr, x, reduce="mean")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): ops.spmm_raw(gr, x, reduce="mean")
    b.record(); torch.cuda.synchronize()
    print("%-60s %.3f ms" % (label, a.elapsed_time(b) / 5))
t(g, "bench graph, real rows")
deg = g.degrees()
for lo, hi in ((0,1),(1,8),(8,16),(16,32),(32,64),(64,128),(128,256),(256,1<<30)):
    m = (deg >= lo) & (deg < hi)
    print("   degree [%d,%d): %.1f %% of rows, %.1f %% of edges" % (lo, hi, 100*m.float().mean(), 100*deg[m].sum()/nnz))
for k in (16, 32, 50, 64, 128, 256):
    rows = nnz // k
    rp = torch.arange(0, rows + 1, device=dev, dtype=torch.int64) * k
    g2 = dgll_amd.CSRGraph(rp, g.col[:rows * k], None, rows, n, check=False)
    t(g2, "same edge stream cut into rows of exactly %d edges (%d rows)" % (k, rows))

# where the time goes by row-length class: each class of rows as its own launch (same X, same column ids)
print("per degree class (rows of that class only, in graph order):")
tot = 0.0
for lo, hi in ((0, 1), (1, 8), (8, 16), (16, 32), (32, 64), (64, 128), (128, 257), (257, 1 << 30)):
    rows = torch.nonzero((deg >= lo) & (deg < hi)).flatten()
    d = deg[rows]
    rp = torch.zeros(rows.numel() + 1, dtype=torch.int64, device=dev)
    torch.cumsum(d, 0, out=rp[1:])
    e = int(rp[-1])
    pos = torch.repeat_interleave(g.rowptr[rows] - rp[:-1], d) + torch.arange(e, device=dev)
    sub = dgll_amd.CSRGraph(rp, g.col[pos].contiguous(), None, rows.numel(), n, check=False)
    sub.plan()
    for _ in range(2): ops.spmm_raw(sub, x, reduce="mean")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): ops.spmm_raw(sub, x, reduce="mean")
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    tot += ms
    print("   degree [%d,%d): %8d rows %10d edges: %.3f ms = %.3f ns/edge, %.1f ns/row" % (lo, hi, rows.numel(), e, ms, ms * 1e6 / max(e, 1), ms * 1e6 / max(rows.numel(), 1)))
print("   sum of the classes: %.3f ms" % tot)
