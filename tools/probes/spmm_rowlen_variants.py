"""Rows of exactly L edges, every column the row's own id (memory out of the picture), F = 47 bf16: the kernel variants side by side."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import dgll_amd
from dgll_amd import _lib, ops

dev = torch.device("cuda:0")
NNZ = 123_718_280
T = _lib.lib.dgll_hip_debug_tune


def med(g, x, reps=6):
    ops.spmm_raw(g, x, reduce="mean"); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.spmm_raw(g, x, reduce="mean"); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


variants = [("default", {}), ("prefetch", {2: 4}), ("flat", {13: 2}), ("row per slot", {5: 2}), ("rpw 4", {1: 4}), ("rpw 8 + prefetch", {1: 8, 2: 4})]
feat = 47
for L in (16, 32, 50, 64, 128):
    n = NNZ // L
    rowptr = torch.arange(0, n + 1, device=dev, dtype=torch.int64) * L
    col = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int32), L)
    x = ops.alloc_features(n, feat, torch.bfloat16, dev, pad_to=64)
    x.normal_()
    row = []
    for name, kw in variants:
        for k in (1, 2, 13, 5):
            T(k, 0)
        for k, v in kw.items():
            T(k, v)
        g = dgll_amd.CSRGraph(rowptr, col, None, n, n, check=False)
        row.append("%s %.3f" % (name, med(g, x)))
    print("F=47 rows of %3d edges: %s" % (L, "   ".join(row)), flush=True)
    del x, col, rowptr
for k in (1, 2, 13, 5):
    T(k, 0)
