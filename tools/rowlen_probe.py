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

# Are the long rows' chunks better processed IN PLACE (at their row's position in the launch) than all at the front of the grid
# (the plan's "chunks first" schedule)?  Emulation without touching the kernel: every row above the threshold is replaced by
# ceil(deg / 256) consecutive pseudo-rows of at most 256 edges (the outputs of the pseudo-rows are partial sums nobody adds up:
# timing only).  The kernel then has no chunk items at all and meets those edges where the row sits.
cuts = []
deg_c = deg.clone()
rp = g.rowptr
long_rows = torch.nonzero(deg > 256).flatten()
pieces = (deg + 255) // 256
pieces = torch.where(deg > 256, pieces, torch.ones_like(pieces))
new_rows = int(pieces.sum())
owner = torch.repeat_interleave(torch.arange(n, device=dev), pieces)                     # real row of every pseudo-row
first = torch.zeros(n + 1, dtype=torch.int64, device=dev)
torch.cumsum(pieces, 0, out=first[1:])
k = torch.arange(new_rows, device=dev) - first[owner]                                    # piece index inside its row
start = rp[owner] + k * 256
end = torch.minimum(start + 256, rp[owner + 1])
rp2 = torch.cat([start, end[-1:]])
assert bool((rp2[1:] >= rp2[:-1]).all()) and int(rp2[-1]) == nnz
g3 = dgll_amd.CSRGraph(rp2.contiguous(), g.col, None, new_rows, n, check=False)
t(g3, "long rows cut IN PLACE into pseudo-rows of <= 256 edges (%d rows, no chunk items)" % new_rows)
t(g, "bench graph, real rows (chunks of long rows scheduled first)")
