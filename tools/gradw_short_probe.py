#!/usr/bin/env python3
"""Split-K weight gradient on SHORT operands (the hops of a sampled mini-batch: 1 k ... 112 k rows): rows per slab.  A slab is one
workgroup per tile type streaming its rows 16 at a time; few slabs leave most CUs idle, many slabs cost a partial (K x N fp32) each
to write and sum."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import dense  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for m in (1024, 11264, 112640, 1_000_000):
    for k1, k2, n in ((256, 256, 256), (90, 0, 256), (256, 0, 41)):
        x1 = torch.randn(m, k1, device=dev).to(torch.bfloat16)
        x1 = dense._as_rows16(x1)
        x2 = dense._as_rows16(torch.randn(m, k2, device=dev).to(torch.bfloat16)) if k2 else None
        g = dense._as_rows16(torch.randn(m, n, device=dev).to(torch.bfloat16))
        line = "M %7d  K %3d+%3d N %3d:" % (m, k1, k2, n)
        ref = None
        for rows in (1024, 512, 256, 128, 64):
            dense._GW_MIN_ROWS = rows
            d = dense._grad_weight_hip(x1, x2, g)[0]
            if ref is None:
                ref = d
            else:
                assert float((d - ref).abs().max()) <= 1e-3 * float(ref.abs().max()) + 1e-3
            line += "  >=%4d rows/slab %6.1f us" % (rows, t(lambda: dense._grad_weight_hip(x1, x2, g)))
        print(line, flush=True)
