#!/usr/bin/env python3
"""Workgroups per CU of the resident-weights MFMA transform at the narrow shapes of the SAGE step (dgll_hip_debug_tune(11, v))."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import _lib, dense, ops  # noqa: E402

dev = torch.device("cuda:0")
M = 2_449_029


def t(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def mat(k, pad=8):
    x = ops.alloc_features(M, k, torch.bfloat16, dev, pad_to=pad)
    x.copy_(torch.randn(M, k, device=dev))
    return x


h256, h100, g47a, g47b, gate = mat(256), mat(100, 64), mat(47, 64), mat(47, 64), mat(256)
w = lambda n, k: (torch.randn(n, k, device=dev) * 0.1).to(torch.bfloat16)
shapes = {
    "256 -> 47": (lambda: dense.transform_bf16(h256, w47_256, ld_align=64), (256 + 47) * 2),
    "256 -> 47 + addend": (lambda: dense.transform_bf16(h256, w47_256, addend=g47a, relu=False, ld_align=64), (256 + 47 + 47) * 2),
    "47 + 47 -> 256 gated": (lambda: dense.transform_bf16(g47a, w256_47, g47b, w256_47b, out_gate=gate), (47 + 47 + 256 + 256) * 2),
    "100 + 100 -> 256": (lambda: dense.transform_bf16(h100, w256_100, h100, w256_100b, relu=True), (200 + 256) * 2),
    "256 + 256 -> 256": (lambda: dense.transform_bf16(h256, w256_256, gate, w256_256b, relu=True), (512 + 256) * 2),
    "256 -> 256": (lambda: dense.transform_bf16(h256, w256_256), (256 + 256) * 2),
}
w47_256, w256_47, w256_47b = w(47, 256), w(256, 47), w(256, 47)
w256_100, w256_100b, w256_256, w256_256b = w(256, 100), w(256, 100), w(256, 256), w(256, 256)
for name, (fn, bytes_per_row) in shapes.items():
    line = "%-24s" % name
    for per_cu in (0, 1, 2, 3, 4):
        _lib.lib.dgll_hip_debug_tune(11, per_cu)
        ms = t(fn)
        line += "  per_cu=%d: %.3f ms %.2f TB/s" % (per_cu, ms, M * bytes_per_row / ms / 1e9)
    print(line, flush=True)
_lib.lib.dgll_hip_debug_tune(11, 0)
