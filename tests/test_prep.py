"""a11 / f4: device-side adjacency preparation vs the reference's scipy pipeline (golden adj_prep_s7)."""
import numpy as np
import torch

from conftest import load_golden
from dgll_amd import prep
from dgll_amd.graph import CSRGraph, as_csr_graph


def test_normalized_adjacency_matches_reference():
    g = load_golden("adj_prep_s7")
    n = g.meta["n"]
    src, dst = g.t("src"), g.t("dst")
    key = torch.unique(src * n + dst)                           # the generator coalesces duplicates to weight 1 first
    graph = prep.normalized_adjacency(torch.div(key, n, rounding_mode="floor"), key % n, n)
    ref = CSRGraph.from_coo(g.t("adj_row"), g.t("adj_col"), g.t("adj_val"), (n, n))
    assert torch.equal(graph.rowptr, ref.rowptr) and torch.equal(graph.col, ref.col)
    np.testing.assert_allclose(graph.val.numpy(), ref.val.numpy(), rtol=1e-6)
    coo = prep.to_torch_coo(graph)
    assert torch.equal(coo._indices()[0], g.t("adj_row")) and torch.equal(coo._indices()[1], g.t("adj_col"))


def test_csr_constructors_agree_and_cache():
    torch.manual_seed(0)
    dense = (torch.rand(40, 40) < 0.1).float()
    a = CSRGraph.from_dense(dense)
    idx = dense.nonzero().t()
    b = CSRGraph.from_coo(idx[0], idx[1], None, dense.shape)
    c = CSRGraph.from_torch_sparse(dense.to_sparse())
    d = CSRGraph.from_torch_sparse(dense.to_sparse_csr())
    for other in (b, c, d):
        assert torch.equal(a.rowptr, other.rowptr) and torch.equal(a.col, other.col)
    # duplicates are summed (PPI/gcn_model.py:56 hands torch an un-coalesced all-ones COO)
    e = CSRGraph.from_edge_index(torch.tensor([[0, 0, 1, 0], [1, 1, 0, 2]]), 3)
    assert e.col.tolist() == [1, 2, 0] and e.val.tolist() == [2.0, 1.0, 1.0]
    # transpose of the transpose is the original; perm maps edge slots
    gt, perm = e.transpose()
    gtt, _ = gt.transpose()
    assert torch.equal(gtt.rowptr, e.rowptr) and torch.equal(gtt.col, e.col) and torch.equal(gtt.val, e.val)
    assert torch.equal(gt.val, e.val[perm])
    # the adjacency cache converts a tensor once
    sp = dense.to_sparse()
    assert as_csr_graph(sp) is as_csr_graph(sp)
    fan = CSRGraph.fixed_fanout(5, 3, "cpu")
    assert fan.rowptr.tolist() == [0, 3, 6, 9, 12, 15] and fan.col.tolist() == list(range(15))
