"""Host-tensor branch of the GAT layers (`_host_heads`, dgll_amd/nn/Convolution/gatconv.py) against the reference's op
ORDER restated densely here -- including a node without any edge, where the two reference layers differ:
gatConv masks with -9e15 and soft-maxes the whole row (gatconv.py:34-36: uniform attention over ALL nodes), sparseGatConv
divides 0 by 0 and asserts (gatconv.py:139-141)."""
import pytest
import torch


def _dense_gatconv(h_in, adj, W, a, alpha, concat):
    """gatconv.py:30-54, dropout inactive."""
    Wh = h_in @ W
    fo = W.shape[1]
    e = torch.nn.functional.leaky_relu(Wh @ a[:fo] + (Wh @ a[fo:]).T, alpha)
    att = torch.softmax(torch.where(adj > 0, e, torch.full_like(e, -9e15)), dim=1)
    hp = att @ Wh
    return torch.nn.functional.elu(hp) if concat else hp


def _adj_with_an_isolated_node(n=40, seed=0):
    g = torch.Generator().manual_seed(seed)
    adj = (torch.rand(n, n, generator=g) < 0.15).float()
    adj = ((adj + adj.T + torch.eye(n)) > 0).float()
    adj[7, :] = 0                                        # node 7 has no out-edges at all (not even the self-loop)
    return adj


@pytest.mark.parametrize("concat", [True, False])
def test_gatconv_on_host_tensors_follows_the_reference_incl_an_edgeless_row(concat):
    from dgll_amd.nn import gatConv

    torch.manual_seed(1)
    adj = _adj_with_an_isolated_node()
    x = torch.randn(adj.shape[0], 9)
    layer = gatConv(9, 5, dropout=0.0, alpha=0.2, concat=concat).eval()
    want = _dense_gatconv(x, adj, layer.W.detach(), layer.a.detach(), 0.2, concat)
    got = layer(x, adj)
    assert torch.isfinite(got).all()
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    # the edgeless row attends uniformly to every node: the mean of Wh (through the activation)
    mean = (x @ layer.W.detach()).mean(0)
    torch.testing.assert_close(got[7], torch.nn.functional.elu(mean) if concat else mean, rtol=1e-5, atol=1e-5)


def test_sparse_gatconv_on_host_tensors_asserts_on_an_edgeless_row_like_the_reference():
    from dgll_amd.nn import sparseGatConv

    torch.manual_seed(2)
    adj = _adj_with_an_isolated_node()
    x = torch.randn(adj.shape[0], 9)
    layer = sparseGatConv(9, 5, dropout=0.0, alpha=0.2).eval()
    with pytest.raises(AssertionError):                  # gatconv.py:141 `assert not torch.isnan(h_prime).any()`
        layer(x, adj)
    adj[7, 7] = 1                                        # with the self-loop every row has an edge
    assert torch.isfinite(layer(x, adj)).all()
