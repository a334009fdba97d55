#!/usr/bin/env python3
"""Kernel-level sweep of dgll_hip_spmm_csr (HIP events, median of reps): edges/s and algorithmic GB/s per shape.
    python tools/spmm_bench.py [--graph products|rmat20|rmat22] [--reps 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import ops, synth  # noqa: E402


def time_ms(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graph", default="products")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--feats", default="256,128,100,64")
    ap.add_argument("--dtypes", default="bf16,f32")
    ap.add_argument("--locality", type=float, default=0.0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.graph == "products":
        g = synth.products_like_graph(dev, seed=0, locality=args.locality)
    else:
        g = synth.rmat_graph(int(args.graph.replace("rmat", "")), 16, seed=0, device=dev, symmetric=False, weighted=False,
                             self_loops=True)
        torch.cuda.empty_cache()
        print("peak memory while building: %.1f GB" % (torch.cuda.max_memory_allocated() / 1e9), flush=True)
    gw = g.with_values(torch.rand(g.nnz, device=dev))
    deg = g.degrees()
    print("graph %s: n=%d nnz=%d avg_deg=%.1f max_deg=%d long_rows=%d" % (args.graph, g.n_rows, g.nnz, g.nnz / g.n_rows,
                                                                         int(deg.max()), g.num_long_rows()), flush=True)
    for dt in args.dtypes.split(","):
        dtype = torch.bfloat16 if dt == "bf16" else torch.float32
        esz = 2 if dt == "bf16" else 4
        for feat in [int(f) for f in args.feats.split(",")]:
            x = ops.alloc_features(g.n_cols, feat, dtype, dev)
            x.copy_(torch.randn(g.n_cols, feat, device=dev).to(dtype))
            for name, graph, weighted in (("unweighted-mean", g, False), ("weighted-sum", gw, True)):
                ms = time_ms(lambda: ops.spmm_raw(graph, x, reduce="mean" if not weighted else "sum"), args.reps)
                b_alg = g.nnz * (feat * esz + 4 + (4 if weighted else 0)) + g.n_rows * (feat * esz + 8)
                print("%-5s F=%-4d %-16s %8.3f ms  %7.2f Gedges/s  %8.1f GB/s alg  (%.1f%% of 8 TB/s)" % (
                    dt, feat, name, ms, g.nnz / ms / 1e6, b_alg / ms / 1e6, b_alg / ms / 1e6 / 80.0), flush=True)


if __name__ == "__main__":
    main()
