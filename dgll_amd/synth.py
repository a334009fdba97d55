"""Synthetic power-law graphs for the benchmarks and the parity tests.

RMAT (a=.57 b=.19 c=.19 d=.05), one random draw per level deciding the (src, dst) bit pair.  Two generators
with the same recursion: numpy (`rmat_edges_np`, bit-reproducible everywhere, used by the tests and the golden
generator) and torch on the device (`rmat_edges_torch`, used for bench-size graphs that are built in HBM).
"""
import numpy as np
import torch

from .graph import CSRGraph

A, B, C = 0.57, 0.19, 0.19


def rmat_edges_np(scale, n_edges, seed, a=A, b=B, c=C):
    rng = np.random.default_rng(seed)
    src = np.zeros(n_edges, dtype=np.int64)
    dst = np.zeros(n_edges, dtype=np.int64)
    for _ in range(scale):
        r = rng.random(n_edges)
        src = (src << 1) | (r >= a + b)
        dst = (dst << 1) | (((r >= a) & (r < a + b)) | (r >= a + b + c))
    return src, dst


def rmat_edges_torch(scale, n_edges, seed, device, a=A, b=B, c=C, chunk=1 << 26):
    """Same recursion on the device, generated in chunks to bound temporary memory."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    srcs, dsts = [], []
    for start in range(0, n_edges, chunk):
        m = min(chunk, n_edges - start)
        src = torch.zeros(m, dtype=torch.int64, device=device)
        dst = torch.zeros(m, dtype=torch.int64, device=device)
        for _ in range(scale):
            r = torch.rand(m, generator=gen, device=device)
            src = (src << 1) | (r >= a + b)
            dst = (dst << 1) | (((r >= a) & (r < a + b)) | (r >= a + b + c))
        srcs.append(src)
        dsts.append(dst)
    return torch.cat(srcs), torch.cat(dsts)


def build_graph(src, dst, n, symmetric=False, self_loops=True, normalise="row", weighted=True):
    """Coalesced CSRGraph from an edge list (torch tensors on any device).

    symmetric: add the reverse of every edge; self_loops: add (i, i); normalise: 'row' gives D^-1 A fp32 weights
    (nn/utils/utils.py:240-247), None gives all-ones; weighted=False drops the value array (SAGE mean/sum)."""
    dev = src.device
    if symmetric:
        src, dst = torch.cat([src, dst]), torch.cat([dst, src])
    if self_loops:
        loops = torch.arange(n, dtype=torch.int64, device=dev)
        src, dst = torch.cat([src, loops]), torch.cat([dst, loops])
    if src.numel() < (1 << 30):
        key = torch.unique(src * n + dst)        # sorted + coalesced (duplicates dropped, weight 1)
        row = torch.div(key, n, rounding_mode="floor")
        col = (key - row * n).to(torch.int32)
        del key
        counts = torch.bincount(row, minlength=n)
    else:
        # torch.unique is limited to < 2^31 elements: coalesce in source-row blocks (RMAT-27 has 2.3e9 edges) and
        # concatenate -- the blocks are disjoint and ascending, so the result is the same sorted edge list
        parts = 4 * (src.numel() >> 30) + 4
        rows, cols, cnts = [], [], []
        step = 1 << 28                 # boolean-mask selection is done on slices (torch's mask indexing overflows past 2^31)
        for p in range(parts):
            lo, hi = (n * p) // parts, (n * (p + 1)) // parts
            keys = []
            for s0 in range(0, src.numel(), step):
                ss, dd = src[s0:s0 + step], dst[s0:s0 + step]
                m = (ss >= lo) & (ss < hi)
                keys.append(ss[m] * n + dd[m])
            key = torch.unique(torch.cat(keys))
            del keys
            r = torch.div(key, n, rounding_mode="floor")
            cnts.append(torch.bincount(r - lo, minlength=hi - lo))
            if weighted:
                rows.append(r)
            cols.append((key - r * n).to(torch.int32))
            del key, r
        del src, dst
        col, counts = torch.cat(cols), torch.cat(cnts)
        row = torch.cat(rows) if weighted else None
        del rows, cols, cnts
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=rowptr[1:])
    val = None
    if weighted:
        if normalise == "row":
            val = (1.0 / counts.clamp(min=1).to(torch.float32))[row]
        else:
            val = torch.ones(col.numel(), dtype=torch.float32, device=dev)
    return CSRGraph(rowptr, col, val, n, n, check=False)


def rmat_graph(scale, edge_factor=16, seed=0, device="cpu", **kw):
    n = 1 << scale
    if torch.device(device).type == "cpu":
        s, d = rmat_edges_np(scale, edge_factor << scale, seed)
        src, dst = torch.from_numpy(s), torch.from_numpy(d)
    else:
        src, dst = rmat_edges_torch(scale, edge_factor << scale, seed, device)
    return build_graph(src, dst, n, **kw)


PRODUCTS_NODES = 2_449_029          # ogbn-products (C3/C4)
PRODUCTS_UNDIRECTED_EDGES = 61_859_140


def products_like_graph(device, seed=0, n=PRODUCTS_NODES, n_undirected=PRODUCTS_UNDIRECTED_EDGES, weighted=False,
                        self_loops=False, locality=0.0, n_blocks=64, exact=False, permute_ids=False):
    """ogbn-products-shaped synthetic graph: N = 2 449 029 nodes, ~61.86 M undirected power-law (RMAT) edges symmetrised
    to ~123.7 M directed ones.

    locality = 0: one global RMAT, ids folded into [0, N) -- no community structure at all (worst case for caches and for
    any partitioner).  locality = p > 0: the node range is cut into `n_blocks` equal communities; every edge picks a
    community, draws its source there (RMAT inside the block), and with probability p its destination in the SAME block,
    else anywhere (global RMAT).  This models the graph after a METIS / community relabelling (BASELINE config 3 names METIS
    partitions; ogbn-products is a co-purchase network with strong communities): a contiguous k-way split then cuts
    about (1 - p)(1 - 1/k) of the edges.

    exact: duplicates coalesce, so one draw of `n_undirected` edges lands several per cent short (dense RMAT blocks: 11 %);
    with exact=True further seeded draws top the edge set up until it holds EXACTLY `n_undirected` distinct undirected
    edges without self-loops, i.e. nnz = 2 * n_undirected (products: 123 718 280).
    permute_ids: relabel the nodes by a seeded random permutation afterwards -- the communities still exist but ids are no
    longer sorted by community, which is what a raw dataset looks like before any reordering."""
    dev = torch.device(device)
    scale = int(np.ceil(np.log2(n)))
    on_cpu = dev.type == "cpu"

    def rmat(sc, m, sd):
        if on_cpu:
            return tuple(torch.from_numpy(x) for x in rmat_edges_np(sc, m, sd))
        return rmat_edges_torch(sc, m, sd, dev)

    def draw(m, sd):
        if locality <= 0.0:
            src, dst = rmat(scale, m, sd)
            return src % n, dst % n
        block = -(-n // n_blocks)
        bscale = int(np.ceil(np.log2(block)))
        gen = torch.Generator(device=dev)
        gen.manual_seed(sd + 7919)
        blk = torch.randint(0, n_blocks, (m,), generator=gen, device=dev)
        local = torch.rand(m, generator=gen, device=dev) < locality
        ls, ld = rmat(bscale, m, sd)               # positions inside the community
        _, gd = rmat(scale, m, sd + 1)             # global destination for the cross-community edges
        base = blk * block
        src = (base + ls % block).clamp(max=n - 1)
        dst = torch.where(local, (base + ld % block).clamp(max=n - 1), gd % n)
        return src, dst

    src, dst = draw(n_undirected, seed)
    if exact:
        def canon(s, d):
            keep = s != d
            s, d = s[keep], d[keep]
            return torch.unique(torch.minimum(s, d) * n + torch.maximum(s, d))

        keys = canon(src, dst)
        del src, dst
        rounds = 0
        while keys.numel() < n_undirected:
            rounds += 1
            if rounds > 64:
                raise RuntimeError("could not reach %d distinct edges (graph too dense for this generator)" % n_undirected)
            need = n_undirected - keys.numel()
            s2, d2 = draw(max(int(need * 1.5) + 1024, 4096), seed + 104729 * rounds)
            fresh = canon(s2, d2)
            del s2, d2
            fresh = fresh[~torch.isin(fresh, keys, assume_unique=True)]
            if fresh.numel() > need:      # keep a seeded random subset of the new edges: exactly `need` of them
                gen = torch.Generator(device=dev)
                gen.manual_seed(seed + 15485863 * rounds)
                fresh = fresh[torch.randperm(fresh.numel(), generator=gen, device=dev)[:need]]
            keys = torch.cat([keys, fresh])
            del fresh
        lo = torch.div(keys, n, rounding_mode="floor")
        hi = keys - lo * n
        del keys
        src, dst = lo, hi
    else:
        keep = src != dst
        src, dst = src[keep], dst[keep]
    if permute_ids:
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed + 2654435761 % (1 << 31))
        perm = torch.randperm(n, generator=gen, device=dev)
        src, dst = perm[src], perm[dst]
    return build_graph(src, dst, n, symmetric=True, self_loops=self_loops, weighted=weighted)
