#!/usr/bin/env python3
"""A/B of the GAT forward pass at the config-4 shape (8 heads x 32 bf16, products-sized graph + self-loops, engine reorder):
dgll_hip_gat_fwd_strided (T[j] gathered per edge: a fifth cache line next to the four of the 512-byte row) against
dgll_hip_gat_fwd_rowscore (t_j = h_j . a2 formed from the gathered row), next to the SpMM at the same width.  Kill criterion of
VERDICT r05 item 3: the row-score forward must come in under 5.1 ms.  Also prints the largest output difference between the forms.

    python tools/gat_rowscore_ab.py [locality] [dtype: bf16|f32]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import ops, ops_edge, synth  # noqa: E402

dev = torch.device("cuda:0")
loc = float(sys.argv[1]) if len(sys.argv) > 1 else 0.9
dtype = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.bfloat16
g = synth.products_like_graph(dev, seed=0, locality=loc, self_loops=True, exact=True, permute_ids=True)
g = g.reorder(method="lpa", seed=0)[0]
g.plan()
heads, fo = 8, 32
n, F = g.n_rows, heads * fo
torch.manual_seed(0)
h = (torch.randn(n, F, device=dev) * 0.5).to(dtype)
a1 = (torch.randn(F, device=dev) * 0.2).to(dtype).float()
a2 = (torch.randn(F, device=dev) * 0.2).to(dtype).float()
s = (h.float() * a1).view(n, heads, fo).sum(2).contiguous()
t = (h.float() * a2).view(n, heads, fo).sum(2).contiguous()


def timeit(fn, reps=10):
    fn()
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


base = timeit(lambda: ops.spmm_raw(g, h, reduce="mean"))
print("spmm mean unweighted, F = %d %s: %.3f ms" % (F, dtype, base))
res = {}
for name, flag in (("T gathered (dgll_hip_gat_fwd_strided)", False), ("t_j from the gathered row (dgll_hip_gat_fwd_rowscore)", True)):
    ops_edge.ROW_SCORES = flag
    out = ops_edge._gat_strided_forward(h, s, t, g, heads, fo, 0.2, 1, False, attn2=a2)
    ms = timeit(lambda: ops_edge._gat_strided_forward(h, s, t, g, heads, fo, 0.2, 1, False, attn2=a2))
    res[flag] = (out[3].float(), out[4])
    print("%-56s %.3f ms = %.2fx the SpMM" % (name, ms, ms / base), flush=True)
d_out = float((res[True][0] - res[False][0]).abs().max())
d_den = float(((res[True][1] - res[False][1]).abs() / res[False][1].abs().clamp(min=1e-30)).max())
print("largest output difference between the forms %.3e (outputs up to %.2f), largest relative denominator difference %.2e" % (
    d_out, float(res[False][0].abs().max()), d_den))
