"""The engine's one-off locality pass (dgll_amd/reorder.py): the relabelled graph is the same graph, results come back in
the caller's node order, and label propagation finds planted communities whose ids were randomly permuted."""
import numpy as np
import pytest
import torch

from dgll_amd import synth
from dgll_amd.graph import CSRGraph


def _dense(g):
    a = torch.zeros(g.n_rows, g.n_cols)
    a[g.row_index(), g.col.long()] = 1.0 if g.val is None else g.val
    return a


@pytest.mark.parametrize("weighted", [False, True])
def test_relabelled_graph_is_isomorphic_and_results_return_in_caller_order(weighted):
    g = synth.products_like_graph("cpu", seed=3, n=600, n_undirected=4000, locality=0.8, n_blocks=6, exact=True,
                                  permute_ids=True, weighted=weighted, self_loops=True)
    g2, perm = g.reorder(method="lpa", seed=1)
    assert sorted(perm.tolist()) == list(range(g.n_rows))                    # a permutation
    assert torch.equal(g2.perm, perm) and torch.equal(g2.inv_perm[perm], torch.arange(g.n_rows))
    a, a2 = _dense(g), _dense(g2)
    assert torch.equal(a2, a[perm][:, perm])                                 # P A P^T, values carried along
    cols = g2.col.long()
    rows = g2.row_index()
    assert bool(((cols[1:] > cols[:-1]) | (rows[1:] != rows[:-1])).all())    # columns ascending inside every row
    x = torch.randn(g.n_rows, 5)
    y = a @ x
    y2 = a2 @ g2.to_engine_order(x)
    torch.testing.assert_close(g2.to_caller_order(y2), y, rtol=1e-5, atol=1e-5)
    # an un-reordered graph passes node data through untouched
    assert g.to_engine_order(x) is x and g.to_caller_order(y) is y


def test_label_propagation_recovers_planted_communities_from_permuted_ids():
    from dgll_amd import reorder

    n, blocks = 20000, 10
    g = synth.products_like_graph("cpu", seed=0, n=n, n_undirected=400_000, locality=0.9, n_blocks=blocks, exact=True,
                                  permute_ids=True)
    labels = reorder.label_propagation(g.rowptr, g.col, n, seed=0)
    same = (labels[g.row_index()] == labels[g.col.long()]).float().mean()
    assert float(same) > 0.85                       # ~90 % of the edges were planted inside a community
    _, sizes = torch.unique(labels, return_counts=True)
    big = sizes[sizes > n // (4 * blocks)]
    assert big.numel() == blocks                    # the ten communities, none merged, none split
    g2, perm = g.reorder(seed=0)
    # neighbours are now close in id space: mean |row - col| falls by far more than half
    before = (g.row_index() - g.col.long()).abs().float().mean()
    after = (g2.row_index() - g2.col.long()).abs().float().mean()
    assert float(after) < 0.35 * float(before)
    # deterministic under the seed
    assert torch.equal(g.reorder(seed=0)[1], perm)


def test_reorder_rejects_rectangular_blocks():
    g = CSRGraph.fixed_fanout(8, 3, "cpu")
    with pytest.raises(ValueError, match="square"):
        g.reorder()


def test_exact_edge_count_and_permutation_of_the_generator():
    g = synth.products_like_graph("cpu", seed=0, n=5000, n_undirected=60_000, locality=0.9, n_blocks=8, exact=True)
    assert g.nnz == 120_000
    gt, _ = g.transpose()
    assert torch.equal(gt.rowptr, g.rowptr) and torch.equal(gt.col, g.col)            # symmetric, no self-loops
    assert int((g.row_index() == g.col.long()).sum()) == 0
    gp = synth.products_like_graph("cpu", seed=0, n=5000, n_undirected=60_000, locality=0.9, n_blocks=8, exact=True, permute_ids=True)
    assert gp.nnz == g.nnz and sorted(gp.degrees().tolist()) == sorted(g.degrees().tolist())
    assert not torch.equal(gp.col, g.col)


@pytest.mark.parametrize("method", ["degree", "random", "lpa"])
def test_reorder_methods_and_the_large_graph_path(method, monkeypatch):
    """'degree' (hubs first), 'random' and the block-wise relabelling used above 2^30 edges (forced here by lowering the
    limit): same graph, results back in caller order (above the limit 'lpa' falls back to 'random')."""
    from dgll_amd import reorder

    g = synth.rmat_graph(10, 8, seed=2, device="cpu", symmetric=False, weighted=True, self_loops=True)
    a = _dense(g)
    x = torch.randn(g.n_rows, 4)
    for large in (False, True):
        if large:
            monkeypatch.setattr(reorder, "_LARGE_NNZ", 0)
        g2, perm = g.reorder(method=method, seed=3)
        assert torch.equal(_dense(g2), a[perm][:, perm])
        torch.testing.assert_close(g2.to_caller_order(_dense(g2) @ g2.to_engine_order(x)), a @ x, rtol=1e-5, atol=1e-5)
        if method == "degree":
            d = g2.degrees()
            assert bool((d[1:] <= d[:-1]).all())                 # hubs first
