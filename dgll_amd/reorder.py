"""One-off locality pass of the aggregation engine: relabel the nodes of a square adjacency so that nodes which share
neighbours get nearby ids, before any SpMM runs.

Why: CSR SpMM at hidden width re-reads feature rows -- counter traffic is 9-18x the compulsory bytes on a graph whose
ids carry no structure (profiles/) -- and how many of those re-reads the 4 MiB L2s and the 256 MiB
Infinity Cache absorb depends only on whether rows that are processed together gather from the same neighbourhood.  The
reference obtains that property from outside: METIS partitions with a relabelled id space (BASELINE config 3;
/root/reference/dgll/GPU Accelerator/utils.py:224-255 reads `node_map` ranges = contiguous ids per part) and a
Leiden-community reordering ("CoG") in its preprocessing scripts.  Here it is a device-side pass of the engine itself:

  * label propagation (semi-synchronous: every node adopts the most frequent label among its neighbours, a seeded random
    half of the nodes per sweep, ties broken by a seeded random priority) -- a handful of sort + segmented-count sweeps over
    the edge list, all torch ops on the device the graph lives on;
  * nodes are then ordered by (community, degree descending, old id): a community's rows become one contiguous range,
    hubs first.

`reorder` returns the relabelled CSRGraph and `perm` with  new row i == old row perm[i]; features / labels are permuted
once with x[perm] and results come back in caller order with y_new[inv_perm] (`CSRGraph.to_caller_order`).
"""
import torch

_LARGE_NNZ = 1 << 30     # above this many edges nothing here sorts the whole edge list in one call


def label_propagation(rowptr, col, n, sweeps=8, seed=0):
    """int64 [n] community label per node (labels are node ids of the community's 'founder')."""
    dev = rowptr.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed + 40503)
    deg = rowptr[1:] - rowptr[:-1]
    row = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    colv = col.long()
    labels = torch.arange(n, device=dev)
    for sweep in range(sweeps):
        prio = torch.randperm(n, generator=gen, device=dev)                 # tie-break priority of every label, this sweep
        key = row * n + labels[colv]
        key, _ = torch.sort(key)
        pair, cnt = torch.unique_consecutive(key, return_counts=True)
        del key
        prow = torch.div(pair, n, rounding_mode="floor")
        plab = pair - prow * n
        del pair
        score = (cnt << 32) | prio[plab]                                     # most frequent label; ties -> highest priority
        best = torch.full((n,), -1, dtype=torch.int64, device=dev)
        best.scatter_reduce_(0, prow, score, "amax", include_self=True)
        del prow, plab, score, cnt
        inv_prio = torch.empty_like(prio)
        inv_prio[prio] = torch.arange(n, device=dev)
        new = torch.where(best >= 0, inv_prio[best.clamp(min=0) & 0xffffffff], labels)
        # semi-synchronous: a fully synchronous sweep oscillates on bipartite-like structure
        move = torch.rand(n, generator=gen, device=dev) < (0.5 if sweep + 1 < sweeps else 1.1)
        changed = int(((new != labels) & move).sum())
        labels = torch.where(move, new, labels)
        if changed < max(n // 1000, 1) and sweep >= 2:
            break
    return labels


def locality_order(rowptr, col, n, method="lpa", seed=0, sweeps=8):
    """perm (int64 [n]): the node that becomes row i of the reordered graph."""
    if method not in ("lpa", "degree", "random"):
        raise ValueError("reorder method must be 'lpa', 'degree' or 'random'")
    deg = rowptr[1:] - rowptr[:-1]
    dmax = int(deg.max()) + 1 if n else 1
    ids = torch.arange(n, device=rowptr.device)
    if method == "random":          # no structure sought: only break whatever correlation the given ids have with degree
        gen = torch.Generator(device=rowptr.device)
        gen.manual_seed(seed + 977)
        return torch.randperm(n, generator=gen, device=rowptr.device)
    if method == "degree":          # hubs first (what "lpa" degenerates to when the graph is one community)
        key = (dmax - 1 - deg) * n + ids
    else:
        labels = label_propagation(rowptr, col, n, sweeps=sweeps, seed=seed)
        # communities largest first (the dust of isolated nodes and tiny components goes to the end); inside a community
        # hubs first, then old id (deterministic)
        _, dense_label, size = torch.unique(labels, return_inverse=True, return_counts=True)
        rank = torch.empty_like(size)
        rank[torch.argsort(size, descending=True, stable=True)] = torch.arange(size.numel(), device=size.device)
        dense_label = rank[dense_label]
        key = (dense_label * dmax + (dmax - 1 - deg)) * n + ids              # < n_comm * dmax * n: fits int64 for any real graph
        if float(dense_label.max() + 1) * dmax * n >= 2.0 ** 62:
            order1 = torch.argsort((dmax - 1 - deg) * n + ids)               # two-pass stable sort instead of one packed key
            return order1[torch.argsort(dense_label[order1], stable=True)]
    return torch.argsort(key)


def relabel(graph, perm):
    """CSRGraph with rows AND columns relabelled: new row i = old row perm[i], new column id = position of the old id in
    perm; columns ascending inside every row."""
    from .graph import CSRGraph

    n = graph.n_rows
    dev = graph.device
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n, device=dev)
    deg = graph.degrees()
    new_deg = deg[perm]
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(new_deg, 0, out=rowptr[1:])
    if graph.nnz >= _LARGE_NNZ:
        # too many edges for one sort (RMAT-27: 2.3e9): move the rows block by block; columns keep their old relative order
        # inside a row (no kernel needs them ascending)
        inv32 = inv.to(torch.int32)
        col = torch.empty(graph.nnz, dtype=torch.int32, device=dev)
        val = None if graph.val is None else torch.empty(graph.nnz, dtype=torch.float32, device=dev)
        block = max(1, int(n * (float(1 << 28) / max(graph.nnz, 1))))           # ~2.7e8 edges per block
        for r0 in range(0, n, block):
            r1 = min(r0 + block, n)
            rows = perm[r0:r1]
            d = deg[rows]
            total = int(d.sum())
            if total == 0:
                continue
            starts = torch.repeat_interleave(graph.rowptr[rows] - (rowptr[r0:r1] - rowptr[r0]), d)
            pos = starts + torch.arange(total, device=dev)
            e0 = int(rowptr[r0])
            col[e0:e0 + total] = inv32[graph.col[pos].long()]
            if val is not None:
                val[e0:e0 + total] = graph.val[pos]
            del starts, pos
        g = CSRGraph(rowptr, col, val, n, n, check=False)
        g.perm, g.inv_perm = perm, inv
        return g
    new_row = inv[graph.row_index()]
    key = new_row * n + inv[graph.col.long()]
    del new_row
    if graph.val is not None:
        key, order = torch.sort(key)
        val = graph.val[order]
        del order
    else:
        key, _ = torch.sort(key)
        val = None
    col = (key - torch.div(key, n, rounding_mode="floor") * n).to(torch.int32)
    g = CSRGraph(rowptr, col, val, n, n, check=False)
    g.perm, g.inv_perm = perm, inv
    return g
