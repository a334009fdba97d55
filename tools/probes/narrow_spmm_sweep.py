"""The narrow SpMM launches of the headline step (F = 47 and F = 100 bf16 on the bench graph): gathers in flight per slot (unroll depth
2 / 4 / 8) x rows per wavefront, interleaved, median of 8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dgll_amd
from dgll_amd import _lib, ops, synth

dev = torch.device("cuda:0")
base = synth.products_like_graph(dev, seed=0, locality=0.9, exact=True, permute_ids=True).reorder(seed=0)[0]


def med(g, x, reps=8):
    ops.spmm_raw(g, x, reduce="mean"); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.spmm_raw(g, x, reduce="mean"); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


for feat in (47, 100, 256):
    x = ops.alloc_features(base.n_cols, feat, torch.bfloat16, dev, pad_to=64)
    x.normal_()
    for rnd in range(2):
        row = []
        for unroll in (0, 2, 4, 8):
            for rpw in (0, 2, 4, 8):
                _lib.lib.dgll_hip_debug_tune(0, unroll); _lib.lib.dgll_hip_debug_tune(1, rpw)
                g = dgll_amd.CSRGraph(base.rowptr, base.col, None, base.n_rows, base.n_cols, check=False)
                row.append("u%d/r%d %.3f" % (unroll, rpw, med(g, x)))
        print("F=%d round %d (0 = the library's choice): %s" % (feat, rnd, "  ".join(row)), flush=True)
_lib.lib.dgll_hip_debug_tune(0, 0); _lib.lib.dgll_hip_debug_tune(1, 0)
