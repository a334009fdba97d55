"""What is the per-edge cost of the SpMM that does not depend on the row width?  The bench graph's row structure with every column index
replaced by (a) 0, (b) the row's own id, (c) a random id inside a 4096-row window around the row: gathers that hit the nearest cache, so
what remains is instruction issue, index traffic and the per-row chain."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dgll_amd
from dgll_amd import ops, synth

dev = torch.device("cuda:0")
base = synth.products_like_graph(dev, seed=0, locality=0.9, exact=True, permute_ids=True).reorder(seed=0)[0]
n, nnz = base.n_rows, base.nnz
rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int32), (base.rowptr[1:] - base.rowptr[:-1]))
gen = torch.Generator(device=dev).manual_seed(0)
cols = {"real": base.col, "all 0": torch.zeros_like(base.col), "own row": rows,
        "window 4096": (rows.long() + torch.randint(-2048, 2048, (nnz,), device=dev, generator=gen)).clamp_(0, n - 1).to(torch.int32),
        "window 65536": (rows.long() + torch.randint(-32768, 32768, (nnz,), device=dev, generator=gen)).clamp_(0, n - 1).to(torch.int32)}


def med(g, x, reps=6):
    ops.spmm_raw(g, x, reduce="mean"); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.spmm_raw(g, x, reduce="mean"); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


for feat in (47, 100, 256):
    x = ops.alloc_features(n, feat, torch.bfloat16, dev, pad_to=64)
    x.normal_()
    print("F=%3d: " % feat + "   ".join("%s %.3f ms" % (name, med(dgll_amd.CSRGraph(base.rowptr, c, None, n, n, check=False), x)) for name, c in cols.items()), flush=True)
