"""Restatement of the reference neighbour sampler (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/dgll/sampling/base_sampler.py:45-58 (sample_neighbours), :30-43 (_subgraph) and
dgllsampler.py:10-21 (multi-hop loop) on plain python lists, using the same stdlib `random.sample` calls in the
same order, so that under `random.seed(s)` the drawn IDs are identical by construction.
"""
import random


def sample_neighbours(adj_lists, nodes, fanout):
    src, dst = [], []
    for v in nodes:
        neighbors = adj_lists[v]
        if len(neighbors) == 0:
            chosen = []
        elif fanout is None or len(neighbors) <= fanout:
            chosen = neighbors
        else:
            chosen = random.sample(neighbors, fanout)       # base_sampler.py:56
        for u in chosen:                                    # base_sampler.py:34-38: src = neighbour, dst = seed
            src.append(u)
            dst.append(v)
    return src, dst


def sample(adj_lists, seed_nodes, fanouts):
    """Returns (input_nodes, output_nodes, [(src, dst) per layer, outermost first])."""
    output_nodes = list(seed_nodes)
    seeds = list(seed_nodes)
    layers = []
    if not fanouts:
        raise UnboundLocalError("input_nodes")               # dgllsampler.py:21 with an empty fan-out list
    for fanout in reversed(fanouts):                         # dgllsampler.py:14
        src, dst = sample_neighbours(adj_lists, seeds, fanout)
        seeds = src                                          # next seeds = src WITH duplicates (dgllsampler.py:17)
        layers.insert(0, (src, dst))
    return seeds, output_nodes, layers
