#!/usr/bin/env python3
"""Forward of the narrowing SAGE layer (256 -> 47, mean) at the bench shape, two ways:
   (a) z = h.Wn; aggz = mean_A(z); out = h.Ws + aggz                     -- h read by two transforms
   (b) [z | s] = h.[Wn | Ws] in one pass; out = s += mean_A(z)            -- dense.transform_bf16_cat + the SpMM's increment form
Usage: python tools/narrow_layer_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dgll_amd import dense, ops, synth  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    g = synth.products_like_graph(dev, seed=0, n=2449029, n_undirected=61859140, locality=0.9, exact=True,
                                  permute_ids=True).reorder(seed=0)[0]
    n = g.n_rows
    h = ops.alloc_features(n, 256, torch.bfloat16, dev)
    h.copy_(torch.relu(torch.randn(n, 256, device=dev)))
    ws, wn = torch.randn(256, 47, device=dev) * 0.06, torch.randn(256, 47, device=dev) * 0.06

    def two():
        z = dense.transform_bf16(h, wn.t(), ld_align=64)
        aggz = ops.spmm_raw(g, z, reduce="mean")
        return dense.transform_bf16(h, ws.t(), addend=aggz, ld_align=64)

    def one():
        z, s = dense.transform_bf16_cat(h, wn.t(), ws.t())
        ops.spmm_raw(g, z, reduce="mean", out=s, accumulate=2)
        return s

    a, b = two(), one()
    err = float((a.float() - b.float()).abs().max()), float(a.float().abs().max())
    print("max |difference| %.4g of max |out| %.4g (one extra bf16 rounding of h.Ws)" % err)
    print("two transforms + SpMM: %.3f ms   one pass + incrementing SpMM: %.3f ms" % (timed(two), timed(one)))
    z64 = dense.transform_bf16(h, wn.t(), ld_align=64)
    z, s = dense.transform_bf16_cat(h, wn.t(), ws.t())
    print("  SpMM F=47 from a 128-byte pitch %.3f ms | from the 256-byte pitch of the pair %.3f | the same incrementing %.3f" % (
        timed(lambda: ops.spmm_raw(g, z64, reduce="mean")), timed(lambda: ops.spmm_raw(g, z, reduce="mean")),
        timed(lambda: ops.spmm_raw(g, z, reduce="mean", out=s, accumulate=2))))
    print("  transforms: 256->47 %.3f, 256->47 + addend %.3f, pair %.3f ms" % (
        timed(lambda: dense.transform_bf16(h, wn.t(), ld_align=64)),
        timed(lambda: dense.transform_bf16(h, ws.t(), addend=z64, ld_align=64)),
        timed(lambda: dense.transform_bf16_cat(h, wn.t(), ws.t()))))


if __name__ == "__main__":
    main()
