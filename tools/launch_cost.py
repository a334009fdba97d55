#!/usr/bin/env python3
"""Host-side cost of individual launches as the mini-batch consumer issues them (diagnostics): python tools/launch_cost.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dgll_amd  # noqa: E402,F401
from dgll_amd import dense, ops, _lib  # noqa: E402

dev = torch.device("cuda:0")
w = torch.randn(602, 256, device=dev)
w2 = torch.randn(256, 256, device=dev)
x = torch.randn(100000, 256, device=dev).to(torch.bfloat16)
g = dgll_amd.CSRGraph.fixed_fanout(10000, 10, dev)


def timeit(name, fn, n=300):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-46s host %.1f us per call, drained after %.1f us per call" % (name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))


for stream in (None, torch.cuda.Stream(dev, priority=-1)):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream(dev))
    with ctx:
        print("stream:", "high priority" if stream is not None else "default")
        timeit("torch.empty((256, 640), bf16)", lambda: torch.empty((256, 640), dtype=torch.bfloat16, device=dev))
        timeit("_pack_now(W[602,256].t())", lambda: dense._pack_now(w.t()))
        timeit("_pack_now(W[256,256].t())", lambda: dense._pack_now(w2.t()))
        timeit("_pack_now(W[256,256])", lambda: dense._pack_now(w2))
        timeit("spmm_raw 10k rows x 10", lambda: ops.spmm_raw(g, x, reduce="mean"))
        timeit("transform_bf16 100k x 256 -> 256", lambda: dense.transform_bf16(x, w2.t()))
        timeit("debug_tune ctypes call (no launch)", lambda: _lib.lib.dgll_hip_debug_tune(7, 0))
