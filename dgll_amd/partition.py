"""Graph-partition ingestion for the multi-GPU path: external partitions -> a relabelled id space + row bounds.

BASELINE config 3 names "METIS partitions".  The reference does not partition in-process; its scripts read what DGL's
`partition_graph` (METIS) wrote: a partition-book JSON with `num_parts` and, per part, a CONTIGUOUS id range in the
relabelled id space -- `node_map["_N"][part] = [start, end]` -- and enumerate a part's nodes as
`torch.arange(start, end)`           (/root/reference/dgll/GPU Accelerator/utils.py:224-255: `load_partition_book`,
`load_partition`, `part2nids`; `metis` is listed in requirements.txt:5 but is not installable here).

This module supplies the same entry points and the two conversions the engine needs:

  * `load_partition_book(path, graph_name)` / `part2nids(gpb, part_id)`   -- the reference's names and behaviour;
  * `bounds_from_book(gpb)`                     -> [0, b1, ..., N] row bounds for `dist.partition_rows / _contiguous`;
  * `relabel_by_parts(part_of_node, n_parts)`   -> (perm, bounds): a raw membership vector (what METIS returns: node -> part)
    becomes the permutation that makes every part contiguous (stable inside a part) plus the bounds;
  * `write_partition_book(...)`                 -- writes the JSON subset read above (round trips, tests, hand-over to the
    reference's scripts);
  * `community_parts(graph, n_parts)`           -- a partitioner of the engine's own for when no external one is given:
    label-propagation communities (dgll_amd/reorder.py) packed into `n_parts` bins of equal edge count, largest first.
    It stands where METIS stands in the reference's pipeline: few cut edges, balanced work.
"""
import json
import os

import torch


def load_partition_book(path, graph_name):
    """utils.py:224-227."""
    with open("{}/{}.json".format(path, graph_name), "r") as f:
        return json.load(f)


def part2nids(gpb, part_id):
    """utils.py:251-253: the node ids of a part are its contiguous range."""
    rng = gpb["node_map"]["_N"][part_id]
    return torch.arange(rng[0], rng[-1], 1)


def bounds_from_book(gpb):
    """Row bounds [0, ..., N] of the book's parts; raises if the ranges are not a contiguous cover in part order."""
    ranges = gpb["node_map"]["_N"]
    if len(ranges) != int(gpb["num_parts"]):
        raise ValueError("partition book lists %d node ranges for num_parts = %d" % (len(ranges), gpb["num_parts"]))
    bounds = [int(ranges[0][0])]
    for p, r in enumerate(ranges):
        if int(r[0]) != bounds[-1] or int(r[-1]) < int(r[0]):
            raise ValueError("part %d covers [%d, %d) but the previous part ended at %d" % (p, r[0], r[-1], bounds[-1]))
        bounds.append(int(r[-1]))
    if bounds[0] != 0:
        raise ValueError("the first part must start at node 0")
    if "num_nodes" in gpb and int(gpb["num_nodes"]) != bounds[-1]:
        raise ValueError("node ranges end at %d but num_nodes = %d" % (bounds[-1], gpb["num_nodes"]))
    return bounds


def write_partition_book(path, graph_name, bounds, num_edges=None, extra=None):
    """The JSON DGL's partition_graph writes, restricted to the keys the reference reads (+ num_nodes / num_edges)."""
    os.makedirs(path, exist_ok=True)
    book = {"graph_name": graph_name, "num_parts": len(bounds) - 1, "num_nodes": int(bounds[-1]),
            "node_map": {"_N": [[int(bounds[p]), int(bounds[p + 1])] for p in range(len(bounds) - 1)]},
            "part_method": "metis", "halo_hops": 1}
    if num_edges is not None:
        book["num_edges"] = int(num_edges)
    if extra:
        book.update(extra)
    for p in range(len(bounds) - 1):
        book["part-%d" % p] = {"part_graph": "part%d/graph.bin" % p}
    with open(os.path.join(path, graph_name + ".json"), "w") as f:
        json.dump(book, f)
    return book


def relabel_by_parts(part_of_node, n_parts=None):
    """Membership vector (node -> part, any integer dtype) -> (perm, bounds): new node i is old node perm[i]; part p owns the
    new ids [bounds[p], bounds[p+1]).  Stable: nodes of a part keep their relative order (DGL's relabelling does the same)."""
    part = part_of_node.to(torch.int64)
    if part.dim() != 1:
        raise ValueError("the membership vector must be 1-D")
    n_parts = (int(part.max()) + 1 if part.numel() else 0) if n_parts is None else int(n_parts)
    if part.numel() and (int(part.min()) < 0 or int(part.max()) >= n_parts):
        raise ValueError("part ids must lie in [0, %d)" % n_parts)
    perm = torch.sort(part, stable=True)[1]
    counts = torch.bincount(part, minlength=n_parts)
    bounds = [0] + torch.cumsum(counts, 0).tolist()
    return perm, bounds


# What one rank's step costs, per unit, on one MI355X (tools/scaling_model.py: every rank of N = 2, 4, 8 of the bench graph run
# through its real kernels, least squares; profiles/r05_scaling_model.log): nanoseconds per own edge (the five SpMM-type passes),
# per HALO ROW (first layer recomputed on it, the transposed pass into it, its share of the weight gradient, the narrow exchange's
# gathers) and per own row (dense transforms, loss).  Equal-EDGE parts leave the rank with the hub communities 25-35 % more halo rows.
# Fit of round 5 over the 14 ranks of N = 2, 4, 8 (two runs: 0.096-0.110 / 1.00-1.03 / 3.5-4.3 ns, constant 0.67 ms per step, largest
# residual 0.08-0.28 ms).  A first guess from round 4's floor analysis (0.125 / 1.56 / 2.3) over-charged the halo rows and moved
# too much off the hub rank (3.56 ms against 4.06-4.16 on the others).
STEP_COST_NS = {"edge": 0.103, "halo_row": 1.0, "row": 3.9}


def part_costs(graph, part, n_parts):
    """Per part: own edges, own rows, HALO rows (distinct remote source nodes its rows gather from), as int64 tensors [n_parts]."""
    deg = graph.degrees()
    dev = deg.device
    edges = torch.zeros(n_parts, dtype=torch.int64, device=dev).index_add_(0, part, deg)
    rows = torch.bincount(part, minlength=n_parts)
    row_part = torch.repeat_interleave(part, deg)
    col = graph.col.long()
    cut = row_part != part[col]
    keys = torch.unique(row_part[cut] * graph.n_cols + col[cut])
    halo = torch.bincount(keys // graph.n_cols, minlength=n_parts)
    return edges, rows, halo


def rebalance_parts(graph, part, n_parts, cost=None, tol=0.01, max_iters=40, log=None):
    """Move boundary nodes from the most expensive part to the cheapest one until every part's MODELLED step cost
    (cost['edge'] . own edges + cost['halo_row'] . halo rows + cost['row'] . own rows) is within `tol` (1 %) of the mean.  The nodes moved
    from p to q are those of p with the most neighbours already in q relative to p (least damage to the cut), lightest first among
    equals; a move's effect on the halo counts is not predictable node by node, so every iteration moves a damped share of the
    difference and the costs are recounted.  Deterministic.  Returns the new node -> part vector."""
    cost = dict(STEP_COST_NS if cost is None else cost)
    part = part.clone()
    if n_parts < 2 or graph.nnz == 0:
        return part
    deg = graph.degrees()
    dev = deg.device
    row = graph.row_index()
    col = graph.col.long()
    n = graph.n_rows
    best, best_spread = part.clone(), None
    for it in range(max_iters):
        edges, rows, halo = part_costs(graph, part, n_parts)
        c = cost["edge"] * edges.double() + cost["halo_row"] * halo.double() + cost["row"] * rows.double()
        mean = float(c.mean())
        spread = float(c.max()) / mean
        if log is not None:
            log.append({"iter": it, "max_over_mean": spread, "min_over_mean": float(c.min()) / mean, "halo_rows": halo.tolist(),
                        "edges": edges.tolist(), "rows": rows.tolist()})
        if best_spread is None or spread < best_spread:
            best, best_spread = part.clone(), spread
        if spread <= 1.0 + tol and float(c.min()) / mean >= 1.0 - tol:
            break
        p, q = int(torch.argmax(c)), int(torch.argmin(c))
        transfer = 0.6 * min(float(c[p]) - mean, mean - float(c[q]))
        if transfer <= 0:
            transfer = 0.3 * (float(c[p]) - float(c[q]))
        in_p = part == p
        pc = part[col]
        rp = in_p[row]
        to_q = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, row[rp], (pc[rp] == q).long())
        to_p = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, row[rp], (pc[rp] == p).long())
        cand = torch.nonzero(in_p).flatten()
        gain = (to_q - to_p)[cand]
        move_cost = cost["edge"] * deg[cand].double() + cost["row"]
        gmin = int(gain.min())
        order = torch.argsort((gain - gmin) * (int(deg.max()) + 2) + (int(deg.max()) + 1 - deg[cand]), descending=True, stable=True)
        cum = torch.cumsum(move_cost[order], 0)
        k = int(torch.searchsorted(cum, torch.tensor(transfer, dtype=cum.dtype, device=dev)))
        if k == 0:
            break
        part[cand[order[:k]]] = q
    return best


def partition_and_order(graph, n_parts, seed=0, sweeps=8, refine=True, imbalance=1.05, stats=None, balance_cost=True):
    """(perm, bounds) for the multi-GPU engine in one pass: parts from `community_parts`, boundary-refined by `refine_parts`
    (refine=False: the packed communities as they come), then balanced on the modelled step cost (`rebalance_parts`: own edges +
    halo rows + own rows, STEP_COST_NS; balance_cost=False keeps the equal-edge parts of round 4), and inside every part the
    locality order of dgll_amd/reorder.py (communities contiguous, largest first; hubs first inside a community).  New node i is
    old node perm[i]; part p owns [bounds[p], bounds[p+1]).  `stats`: a dict that receives cut / balance before and after."""
    part, dense = community_parts(graph, n_parts, seed=seed, sweeps=sweeps, return_communities=True)
    if stats is not None:
        stats["before"] = partition_quality(graph, part, n_parts)
    if refine and n_parts > 1:
        part = refine_parts(graph, part, n_parts, imbalance=imbalance, seed=seed, log=None if stats is None else stats.setdefault("passes", []))
        if stats is not None:
            stats["after"] = partition_quality(graph, part, n_parts)
    if balance_cost and n_parts > 1:
        part = rebalance_parts(graph, part, n_parts, log=None if stats is None else stats.setdefault("rebalance", []))
        if stats is not None:
            stats["balanced"] = partition_quality(graph, part, n_parts)
    n = graph.n_rows
    deg = graph.degrees()
    dmax = int(deg.max()) + 1 if n else 1
    ids = torch.arange(n, device=deg.device)
    # three stable sorts, least significant key first: (degree descending, id) -> community -> part
    order = torch.argsort((dmax - 1 - deg) * n + ids)
    order = order[torch.argsort(dense[order], stable=True)]
    order = order[torch.argsort(part[order], stable=True)]
    counts = torch.bincount(part, minlength=n_parts)
    return order, [0] + torch.cumsum(counts, 0).tolist()


def community_parts(graph, n_parts, seed=0, sweeps=8, return_communities=False):
    """node -> part for a square CSRGraph: communities found by label propagation, packed largest-first into the part
    with the fewest edges so far (longest-processing-time bin packing on the communities' edge counts).  Communities larger
    than a fair share are split by node order so that no part exceeds it by more than one community's worth."""
    from . import reorder

    if graph.n_rows != graph.n_cols:
        raise ValueError("partitioning needs a square adjacency")
    n = graph.n_rows
    labels = reorder.label_propagation(graph.rowptr, graph.col, n, sweeps=sweeps, seed=seed)
    _, dense, size = torch.unique(labels, return_inverse=True, return_counts=True)
    rank = torch.empty_like(size)                    # communities numbered largest first (as reorder.locality_order)
    rank[torch.argsort(size, descending=True, stable=True)] = torch.arange(size.numel(), device=size.device)
    dense = rank[dense]
    deg = graph.degrees()
    n_comm = int(dense.max()) + 1 if n else 0
    work = torch.zeros(n_comm, dtype=torch.int64, device=deg.device).index_add_(0, dense, deg + 1)   # +1: isolated nodes count
    order = torch.argsort(work, descending=True, stable=True).tolist()
    work_l = work.tolist()
    fair = (sum(work_l) + n_parts - 1) // max(n_parts, 1)
    load = [0] * n_parts
    part_of_comm = [0] * n_comm
    split = []                                       # communities bigger than a fair share: dealt out node by node below
    for c in order:
        if work_l[c] > fair:
            split.append(c)
            continue
        p = min(range(n_parts), key=lambda q: (load[q], q))
        part_of_comm[c] = p
        load[p] += work_l[c]
    part = torch.tensor(part_of_comm, dtype=torch.int64, device=deg.device)[dense]
    for c in split:                                  # cut an oversized community into runs that top the lightest parts up
        nodes = torch.nonzero(dense == c).flatten()
        w = (deg[nodes] + 1).cumsum(0)
        start, done = 0, 0
        while start < nodes.numel():
            p = min(range(n_parts), key=lambda q: (load[q], q))
            room = max(fair - load[p], (work_l[c] + n_parts - 1) // n_parts)
            end = int(torch.searchsorted(w, torch.tensor(done + room, device=w.device), right=True))
            end = min(max(end, start + 1), nodes.numel())
            part[nodes[start:end]] = p
            took = int(w[end - 1]) - done
            load[p] += took
            done += took
            start = end
    return (part, dense) if return_communities else part


def partition_quality(graph, part, n_parts):
    """{'cut': share of the directed edges whose endpoints lie in different parts, 'balance': edges of the heaviest part over
    the mean, 'rows': nodes per part, 'edges': edges per part}."""
    deg = graph.degrees()
    row_part = torch.repeat_interleave(part, deg)
    cut = int((row_part != part[graph.col.long()]).sum())
    edges = torch.zeros(n_parts, dtype=torch.int64, device=deg.device).index_add_(0, part, deg)
    rows = torch.bincount(part, minlength=n_parts)
    mean = max(float(edges.sum()) / max(n_parts, 1), 1.0)
    return {"cut": cut / max(graph.nnz, 1), "cut_edges": cut, "balance": float(edges.max()) / mean,
            "rows": rows.tolist(), "edges": edges.tolist()}


def refine_parts(graph, part, n_parts, imbalance=1.05, passes=24, seed=0, log=None):
    """Boundary refinement of a node -> part vector (the step METIS runs after its coarse partition; here after
    `community_parts`, whose label-propagation communities are packed as they come: on the bench graph it leaves 13.1 % of the
    edges cut where the generator's floor is 8.75 %).

    Majority vote under a balance cap, on the device, all torch ops: every pass counts each node's neighbours per part
    (one index_add over the edge list), proposes for every node the part that holds most of them, and moves the nodes with
    a positive gain (neighbours in the target part minus neighbours in the own one) -- a seeded random half of them per
    pass while the partition still changes a lot (two adjacent nodes swapping sides in the same pass would oscillate), all of
    them at the end -- best gain first, as long as the target part stays under `imbalance` x the mean work (work of a node =
    degree + 1, as in community_parts).  Stops when a pass improves the cut by less than 0.1 % of the edges.
    Deterministic for a given seed."""
    n = graph.n_rows
    dev = graph.device
    part = part.clone()
    if n == 0 or n_parts < 2 or graph.nnz == 0:
        return part
    deg = graph.degrees()
    work = deg + 1
    cap = int(imbalance * float(work.sum()) / n_parts) + 1
    row = graph.row_index()
    col = graph.col.long()
    ones = torch.ones(graph.nnz, dtype=torch.int32, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed + 611953)
    ids = torch.arange(n, device=dev)
    prev_cut, half = None, True
    best_cut, best_part = None, None
    for it in range(passes):
        cnt = torch.zeros(n * n_parts, dtype=torch.int32, device=dev)
        cnt.index_add_(0, row * n_parts + part[col], ones)
        cnt = cnt.view(n, n_parts)
        own = cnt.gather(1, part.unsqueeze(1)).squeeze(1)
        cut = int(deg.sum()) - int(own.sum())
        if log is not None:
            log.append({"pass": it, "cut": cut / graph.nnz, "half": half})
        if best_cut is None or cut < best_cut:
            best_cut, best_part = cut, part.clone()
        if prev_cut is not None:
            improved = prev_cut - cut
            if half and improved < graph.nnz // 200:
                half = False                      # the partition has settled: move every node that gains
            elif not half and improved < graph.nnz // 1000:
                break
        prev_cut = cut
        other = cnt.scatter(1, part.unsqueeze(1), -1)
        best_cnt, best = other.max(1)
        gain = (best_cnt - own).to(torch.int64)
        del cnt, other
        cand = gain > 0
        if half:
            cand &= torch.rand(n, generator=gen, device=dev) < 0.5
        cidx = ids[cand]
        if cidx.numel() == 0:
            if not half:
                break
            half = False
            continue
        # best gain first inside every target part; accept the prefix that fits under the cap
        g_c, b_c, w_c = gain[cidx], best[cidx], work[cidx]
        gmax = int(g_c.max()) + 1
        order = torch.argsort(b_c * gmax + (gmax - 1 - g_c), stable=True)
        cidx, b_c, w_c = cidx[order], b_c[order], w_c[order]
        load = torch.zeros(n_parts, dtype=torch.int64, device=dev).index_add_(0, part, work)
        csum = torch.cumsum(w_c, 0)
        seg_cnt = torch.bincount(b_c, minlength=n_parts)
        seg_start = torch.cumsum(seg_cnt, 0) - seg_cnt
        base = torch.where(seg_start > 0, csum[(seg_start - 1).clamp(min=0)], torch.zeros_like(seg_start))
        base = torch.where(seg_cnt > 0, base, torch.zeros_like(base))
        within = csum - base[b_c]
        ok = within <= (cap - load)[b_c]
        part[cidx[ok]] = b_c[ok]
    return best_part if best_part is not None else part
