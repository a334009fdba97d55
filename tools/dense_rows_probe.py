#!/usr/bin/env python3
"""ns per row of the MFMA transform shapes of the GraphSAGE step at the row counts a rank of N = 8 / 4 / 1 owns:
the persistent resident-weights kernel (weights of the whole reduction in LDS, one or two workgroups per CU) against the 4-wave
kernel (dgll_hip_debug_tune(4, 1)), on random operands.  HIP events, 20 launches each after 3 warm-ups.

    python tools/dense_rows_probe.py [rows,rows,...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import _lib, dense, ops  # noqa: E402

dev = torch.device("cuda:0")
rows_list = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [254_145, 313_000, 562_000, 941_000, 2_449_029]
bf = torch.bfloat16


def feats(m, k, pad=8):
    x = ops.alloc_features(m, k, bf, dev, pad_to=pad)
    x.copy_(torch.randn(m, k, device=dev) * 0.3)
    return x


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


shapes = [
    ("100+100->256 relu, sign bits", 100, 100, 256, dict(relu=True, bits_out=True)),
    ("256+256->256 relu, sign bits", 256, 256, 256, dict(relu=True, bits_out=True)),
    ("256->47", 256, 0, 47, dict(ld_align=64)),
    ("256->47 + addend", 256, 0, 47, dict(relu=True, addend=True)),
    ("47+47->256 gate bits", 47, 47, 256, dict(gate=True)),
    ("256+256->256 gate bits", 256, 256, 256, dict(gate=True)),
    ("256->256 gate bits (halo rows)", 256, 0, 256, dict(gate=True)),
]
print("%-34s %10s | %s" % ("shape", "rows", "resident-weights kernel: ms, ns/row   |   4-wave kernel: ms, ns/row"))
for name, k1, k2, n, opt in shapes:
    w1 = (torch.randn(n, k1, device=dev) * 0.1)
    w2 = (torch.randn(n, k2, device=dev) * 0.1) if k2 else None
    for m in rows_list:
        a1 = feats(m, k1, 64 if k1 == 100 else 8)
        a2 = feats(m, k2) if k2 else None
        kw = {}
        if opt.get("relu"):
            kw["relu"] = True
        if opt.get("bits_out"):
            kw["bits_out"] = True
        if opt.get("ld_align"):
            kw["ld_align"] = opt["ld_align"]
        if opt.get("addend"):
            kw["addend"] = feats(m, n, 64)
        if opt.get("gate"):
            gate = feats(m, n)
            kw["out_gate"] = gate
            kw["gate_bits"] = (torch.randint(-2 ** 31, 2 ** 31 - 1, (m, dense.bit_words(n)), device=dev, dtype=torch.int64)).to(torch.int32)
        res = []
        for knob in (0, 1):
            _lib.check(_lib.lib.dgll_hip_debug_tune(4, knob), "tune")
            ms = timeit(lambda: dense.transform_bf16(a1, w1, a2, w2, **kw))
            res.append(ms)
        _lib.check(_lib.lib.dgll_hip_debug_tune(4, 0), "tune")
        print("%-34s %10d | %.4f ms %.3f ns/row   |   %.4f ms %.3f ns/row   %s" % (
            name, m, res[0], res[0] * 1e6 / m, res[1], res[1] * 1e6 / m, "<- 4-wave faster" if res[1] < 0.97 * res[0] else ""), flush=True)
        del a1, a2, kw
        torch.cuda.empty_cache()
