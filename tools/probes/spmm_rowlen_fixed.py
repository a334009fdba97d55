"""Per-row against per-edge cost of the SpMM with the memory system out of the picture: 123.7 M edges in rows of EXACTLY L edges, every
column index the row's own id (gathers hit the nearest cache), F = 47 / 256 bf16.  Per-row cost shows as time growing with the row count."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dgll_amd
from dgll_amd import ops

dev = torch.device("cuda:0")
NNZ = 123_718_280


def med(g, x, reps=6):
    ops.spmm_raw(g, x, reduce="mean"); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.spmm_raw(g, x, reduce="mean"); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


for feat in (47, 256):
    for L in (8, 16, 32, 50, 64, 128, 256):
        n = NNZ // L
        rowptr = torch.arange(0, n + 1, device=dev, dtype=torch.int64) * L
        col = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int32), L)
        x = ops.alloc_features(n, feat, torch.bfloat16, dev, pad_to=64)
        x.normal_()
        g = dgll_amd.CSRGraph(rowptr, col, None, n, n, check=False)
        t = med(g, x)
        print("F=%3d rows of %3d edges (%8d rows): %.3f ms = %.1f ps per edge, %.2f ns per row" % (feat, L, n, t, t * 1e9 / (n * L), t * 1e6 / n), flush=True)
        del g, x, col, rowptr
