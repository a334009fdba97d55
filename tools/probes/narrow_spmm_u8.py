import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, dgll_amd
from dgll_amd import _lib, ops, synth
dev = torch.device("cuda:0")
base = synth.products_like_graph(dev, seed=0, locality=0.9, exact=True, permute_ids=True).reorder(seed=0)[0]
T = _lib.lib.dgll_hip_debug_tune
def med(g, x, val=None, reps=8):
    kw = dict(reduce="mean") if val is None else dict(val=val, reduce="sum")
    ops.spmm_raw(g, x, **kw); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.spmm_raw(g, x, **kw); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
val = torch.rand(base.nnz, device=dev)
for feat in (47, 100):
    x = ops.alloc_features(base.n_cols, feat, torch.bfloat16, dev, pad_to=64); x.normal_()
    for rnd in range(2):
        row = []
        for u in (4, 8):
            T(0, u)
            g = dgll_amd.CSRGraph(base.rowptr, base.col, None, base.n_rows, base.n_cols, check=False)
            row.append("U=%d unweighted %.3f weighted %.3f" % (u, med(g, x), med(g, x, val)))
        print("F=%d round %d: %s" % (feat, rnd, "   ".join(row)), flush=True)
T(0, 4)
