#!/usr/bin/env python3
"""A/B of second-level row orders INSIDE the label-propagation communities of the bench graph (products-sized, ids permuted,
locality 0.9): does any order of a community's rows make consecutive rows share more neighbours than "hubs first" does?
    python tools/second_level_probe.py [feat] [locality]
Orders (all keep the communities contiguous, largest first):
    hubs     degree descending (the engine's order)
    bfs      breadth-first level from the community's hub over intra-community edges, then degree
    lpa2     a second label propagation on the intra-community edges only (sub-communities, largest first), then degree
    hubsig   rows keyed by their most-connected neighbour (its position in the hubs order), then degree
    minhash  rows keyed by the minimum of a seeded hash over their neighbour set (similar sets -> equal keys), then degree
Each is timed with the default block -> row mapping and with XCD-contiguous rows (debug flag 1: one XCD's workgroups walk one
contiguous window of rows)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import _lib, ops, reorder, synth  # noqa: E402

dry = os.environ.get("DGLL_PROBE_DRY") == "1"          # host dry run of the orderings on a small graph (no timing)
dev = torch.device("cpu" if dry else "cuda:0")
feat = int(sys.argv[1]) if len(sys.argv) > 1 else 256
loc = float(sys.argv[2]) if len(sys.argv) > 2 else 0.9

g0 = (synth.products_like_graph(dev, seed=0, locality=loc, n=40000, n_undirected=400000, n_blocks=8, permute_ids=True) if dry else
      synth.products_like_graph(dev, seed=0, locality=loc, exact=True, permute_ids=True))
n = g0.n_rows
ids = torch.arange(n, device=dev)
deg = g0.degrees()
dmax = int(deg.max()) + 1
labels = reorder.label_propagation(g0.rowptr, g0.col, n, seed=0)
_, comm, size = torch.unique(labels, return_inverse=True, return_counts=True)
rank = torch.empty_like(size)
rank[torch.argsort(size, descending=True, stable=True)] = torch.arange(size.numel(), device=dev)
comm = rank[comm]
n_comm = int(size.numel())
print("graph: %d nodes %d edges, %d communities, largest %d, >1000 nodes: %d" %
      (n, g0.nnz, n_comm, int(size.max()), int((size > 1000).sum())), flush=True)
row = g0.row_index()
col = g0.col.long()
intra = comm[row] == comm[col]
print("intra-community edges: %.1f %%" % (100.0 * float(intra.float().mean())), flush=True)


def order_by(second):
    """perm for key (community, second, degree descending, id); `second` int64 >= 0."""
    o = torch.argsort((dmax - 1 - deg) * n + ids)
    o = o[torch.argsort(second[o], stable=True)]
    return o[torch.argsort(comm[o], stable=True)]


def second_bfs():
    level = torch.full((n,), 1 << 20, dtype=torch.int64, device=dev)
    # the hub of every community: its highest-degree node
    hub_key = torch.zeros(n_comm, dtype=torch.int64, device=dev)
    hub_key.scatter_reduce_(0, comm, deg * n + ids, "amax", include_self=True)
    level[hub_key % n] = 0
    r, c = row[intra], col[intra]
    for _ in range(12):
        cand = level[c] + 1
        new = level.clone()
        new.scatter_reduce_(0, r, cand, "amin", include_self=True)
        if bool((new == level).all()):
            break
        level = new
    hist = torch.bincount(level.clamp(max=12))
    print("  bfs levels (0, 1, 2, ...):", hist.tolist(), flush=True)
    return level.clamp(max=12)


def second_lpa2():
    r, c = row[intra], col[intra]
    rp = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.bincount(r, minlength=n), 0, out=rp[1:])
    sub = reorder.label_propagation(rp, c.to(torch.int32), n, sweeps=5, seed=1)
    _, dense, ssz = torch.unique(sub, return_inverse=True, return_counts=True)
    srank = torch.empty_like(ssz)
    srank[torch.argsort(ssz, descending=True, stable=True)] = torch.arange(ssz.numel(), device=dev)
    print("  second-level labels: %d (first level %d); largest %d" % (int(ssz.numel()), n_comm, int(ssz.max())), flush=True)
    return srank[dense]


def second_hubsig(pos_in_hubs_order):
    key = torch.full((n,), n, dtype=torch.int64, device=dev)
    key.scatter_reduce_(0, row, pos_in_hubs_order[col], "amin", include_self=True)
    return key


def second_minhash():
    h = (col * 2654435761 + 40503) % 2147483647
    key = torch.full((n,), 1 << 31, dtype=torch.int64, device=dev)
    key.scatter_reduce_(0, row, h, "amin", include_self=True)
    return key


def time_pass(g, x, flags, reps=7):
    if dry:
        return 0.0
    _lib.check(_lib.lib.dgll_hip_debug_tune(2, flags), "tune")
    ops.spmm_raw(g, x, reduce="mean")
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.spmm_raw(g, x, reduce="mean"); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    _lib.check(_lib.lib.dgll_hip_debug_tune(2, 0), "tune")
    ts.sort()
    return ts[len(ts) // 2]


hubs_perm = order_by(torch.zeros(n, dtype=torch.int64, device=dev))
pos = torch.empty_like(hubs_perm)
pos[hubs_perm] = ids
variants = [("hubs", lambda: hubs_perm), ("bfs", lambda: order_by(second_bfs())), ("lpa2", lambda: order_by(second_lpa2())),
            ("hubsig", lambda: order_by(second_hubsig(pos))), ("minhash", lambda: order_by(second_minhash()))]
for name, make in variants:
    perm = make()
    g = reorder.relabel(g0, perm)
    x = None if dry else ops.alloc_features(n, feat, torch.bfloat16, dev)
    if not dry:
        x.copy_(torch.randn(n, feat, device=dev))
    t0, t1 = time_pass(g, x, 0), time_pass(g, x, 1)
    print("%-8s F=%d mean pass: %.3f ms default mapping, %.3f ms XCD-contiguous rows" % (name, feat, t0, t1), flush=True)
    del g, x, perm
    if not dry:
        torch.cuda.empty_cache()
