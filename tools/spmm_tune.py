#!/usr/bin/env python3
"""Interleaved A/B of the SpMM tuning knobs on the headline shape (products-sized graph, F=256 bf16, mean).
    python tools/spmm_tune.py [feat] [locality] [raw|lpa|lpa-id|sorted]
graph variants: raw = permuted ids, no reordering; lpa = the engine's reorder (communities, hubs first);
lpa-id = communities, old-id order inside; sorted = the generator's community-sorted ids."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dgll_amd  # noqa: E402
from dgll_amd import _lib, ops, reorder, synth  # noqa: E402

dev = torch.device("cuda:0")
feat = int(sys.argv[1]) if len(sys.argv) > 1 else 256
loc = float(sys.argv[2]) if len(sys.argv) > 2 else 0.9
variants = (sys.argv[3] if len(sys.argv) > 3 else "lpa").split(",")


def build(variant):
    if variant == "sorted":
        return synth.products_like_graph(dev, seed=0, locality=loc, exact=True)
    g = synth.products_like_graph(dev, seed=0, locality=loc, exact=True, permute_ids=True)
    if variant == "raw":
        return g
    if variant == "lpa":
        return g.reorder(seed=0)[0]
    if variant == "lpa-id":
        labels = reorder.label_propagation(g.rowptr, g.col, g.n_rows, seed=0)
        _, dense, size = torch.unique(labels, return_inverse=True, return_counts=True)
        rank = torch.empty_like(size)
        rank[torch.argsort(size, descending=True, stable=True)] = torch.arange(size.numel(), device=dev)
        perm = torch.argsort(rank[dense] * g.n_rows + torch.arange(g.n_rows, device=dev))
        return reorder.relabel(g, perm)
    raise SystemExit("unknown variant " + variant)


def tune(**kw):
    keys = {"unroll": 0, "rpw": 1, "flags": 2, "threshold": 3}
    for k, v in kw.items():
        _lib.check(_lib.lib.dgll_hip_debug_tune(keys[k], v), "tune")


def run(g, x, weighted, reps=6):
    val = torch.rand(g.nnz, device=dev) if weighted else None
    out = ops.alloc_features(g.n_rows, feat, torch.bfloat16, dev)
    out.zero_()
    kw = dict(val=val, reduce="sum", out=out, accumulate=True, gate=x) if weighted else dict(reduce="mean")
    ops.spmm_raw(g, x, **kw)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.spmm_raw(g, x, **kw); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


thresholds = [int(v) for v in os.environ.get("TUNE_THRESHOLDS", "128,256").split(",")]
rpws = [int(v) for v in os.environ.get("TUNE_RPW", "0,2,8").split(",")]
remaps = [int(v) for v in os.environ.get("TUNE_REMAP", "0,1").split(",")]
configs = [("thr%d rpw%d remap%d" % (t, r, f), dict(unroll=4, rpw=r, flags=f, threshold=t))
           for t in thresholds for r in rpws for f in remaps]
for variant in variants:
    base = build(variant)
    x = torch.randn(base.n_cols, feat, device=dev).to(torch.bfloat16)
    print("== graph variant %s: nnz %d, max degree %d" % (variant, base.nnz, int(base.degrees().max())), flush=True)
    for rnd in range(2):
        for name, kw in configs:
            tune(**kw)
            g = dgll_amd.CSRGraph(base.rowptr, base.col, None, base.n_rows, base.n_cols, check=False)
            med, mn = run(g, x, False)
            medw, mnw = run(g, x, True) if rnd == 0 else (0, 0)
            print("round %d  %-24s fwd median %.3f ms  min %.3f ms  (%.2f Gedges/s)   bwd-style weighted+accumulate+gate median %.3f" % (
                rnd, name, med, mn, base.nnz / med / 1e6, medw), flush=True)
    del base, x
