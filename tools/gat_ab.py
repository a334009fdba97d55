#!/usr/bin/env python3
"""Per-pass timing of the three GAT gather passes next to the SpMM at the same width on the same graph (config 4 shape:
8 heads x 32 = 256 columns bf16, products-sized graph + self-loops, engine reorder).  HIP events, `reps` launches each.

    python tools/gat_ab.py [locality] [probe-list]     probe-list: comma-separated dgll_hip_debug_tune(8, v) values
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import _lib, ops, ops_edge, synth  # noqa: E402

dev = torch.device("cuda:0")
loc = float(sys.argv[1]) if len(sys.argv) > 1 else 0.9
probes = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
unrolls = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2]
g = synth.products_like_graph(dev, seed=0, locality=loc, self_loops=True, exact=True, permute_ids=True)
g = g.reorder(method="lpa", seed=0)[0]
gt, perm = g.transpose()
g.plan(), gt.plan()
heads, fo = 8, 32
n, F = g.n_rows, heads * fo
torch.manual_seed(0)
h = torch.randn(n, F, device=dev).to(torch.bfloat16)
s = torch.randn(n, heads, device=dev)
t = torch.randn(n, heads, device=dev)
go = torch.randn(n, F, device=dev).to(torch.bfloat16)
w = torch.rand(g.nnz, device=dev)


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


out = torch.empty_like(h)
rowsum = torch.empty(n, heads, device=dev)
dn = torch.empty_like(h)
dd = torch.empty(n, heads, device=dev)
gs = torch.empty(n, heads, device=dev)
gh = torch.empty_like(h)
gtt = torch.empty(n, heads, device=dev)

ref = {
    "spmm mean unweighted A": timeit(lambda: ops.spmm_raw(g, h, reduce="mean")),
    "spmm sum weighted A^T": timeit(lambda: ops.spmm_raw(gt, h, val=w, reduce="sum")),
}
for k, v in ref.items():
    print("%-28s %.3f ms" % (k, v), flush=True)
b_alg = g.nnz * (F * 2 + 4 + 4 * heads) + n * (F * 2 + 8 + 8 * heads)
gens = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]
sd = torch.empty(n, 2 * heads, device=dev)
for gen in gens:
    _lib.lib.dgll_hip_debug_tune(9, gen)
    f = timeit(lambda: ops_edge.gat_fwd_part(g, h, s, t, out, rowsum, heads, fo, 0.2, 1, 0, 0))
    r = timeit(lambda: ops_edge.gat_bwd_rows_part(g, h, s, t, out, go, rowsum, dn, dd, gs, heads, fo, 0.2, 1, 0))
    c = timeit(lambda: ops_edge.gat_bwd_cols_part(gt, dn, h, t, s, dd, gh, gtt, heads, fo, 0.2))
    base = ref["spmm mean unweighted A"]
    print("gen %d: fwd %.3f ms (%.2fx spmm, alg %.0f GB/s = %.2f of 8 TB/s) | bwd_rows %.3f ms (%.2fx) | "
          "bwd_cols (separate S, DD arrays) %.3f ms (%.2fx)" % (gen, f, f / base, b_alg / f / 1e6, b_alg / f / 1e6 / 8000, r, r / base,
                                                               c, c / base), flush=True)
    if gen == 0:   # both backward passes through the strided entry point ({s, dd} side by side): cols = total - rows
        ws_bytes = max(int(_lib.lib.dgll_hip_gat_workspace_bytes(g.plan(), heads, fo)),
                       int(_lib.lib.dgll_hip_gat_workspace_bytes(gt.plan(), heads, fo)))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream

        def both():
            code = _lib.lib.dgll_hip_gat_bwd_strided(
                st, g.plan(), gt.plan(), g.rowptr.data_ptr(), g.col.data_ptr(), gt.rowptr.data_ptr(), gt.col.data_ptr(),
                h.data_ptr(), h.stride(0), s.data_ptr(), t.data_ptr(), heads, t.data_ptr(), out.data_ptr(), out.stride(0),
                go.data_ptr(), go.stride(0), _lib.BF16, rowsum.data_ptr(), dn.data_ptr(), dn.stride(0), sd.data_ptr(), 2 * heads,
                gh.data_ptr(), gh.stride(0), gs.data_ptr(), gtt.data_ptr(), n, n, heads, fo, 0.2, 1, ws.data_ptr(), ws_bytes)
            _lib.check(code, "dgll_hip_gat_bwd_strided")

        bt = timeit(both)
        print("gen 0: both backward passes, {s, dd} side by side: %.3f ms -> bwd_cols %.3f ms (%.2fx)" % (bt, bt - r, (bt - r) / base),
              flush=True)
_lib.lib.dgll_hip_debug_tune(9, 0)
