"""The narrow SpMM launches of the headline step (F = 47, 100; 256 for reference) in the kernel variants the library holds: default
(wave per row), next-row index prefetch (flags 4), flattened edge stream (tune 13 = 2 forces it, 1 forbids), row per slot (tune 5 = 2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dgll_amd
from dgll_amd import _lib, ops, synth

dev = torch.device("cuda:0")
base = synth.products_like_graph(dev, seed=0, locality=0.9, exact=True, permute_ids=True).reorder(seed=0)[0]
T = _lib.lib.dgll_hip_debug_tune


def med(g, x, reps=8):
    ops.spmm_raw(g, x, reduce="mean"); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.spmm_raw(g, x, reduce="mean"); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


variants = [("default", {}), ("prefetch", {2: 4}), ("flat forced", {13: 2}), ("flat off", {13: 1}), ("row per slot", {5: 2})]
for feat in (47, 100, 256):
    x = ops.alloc_features(base.n_cols, feat, torch.bfloat16, dev, pad_to=64)
    x.normal_()
    ref = None
    for rnd in range(2):
        row = []
        for name, kw in variants:
            for k in (2, 13, 5):
                T(k, 0)
            for k, v in kw.items():
                T(k, v)
            g = dgll_amd.CSRGraph(base.rowptr, base.col, None, base.n_rows, base.n_cols, check=False)
            row.append("%s %.3f" % (name, med(g, x)))
        print("F=%d round %d: %s" % (feat, rnd, "   ".join(row)), flush=True)
for k in (2, 13, 5):
    T(k, 0)
