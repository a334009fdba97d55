"""dgll.nn layers vs the golden vectors produced by the imported reference (tests/golden/gen_goldens.py).

Every case runs twice: on CPU tensors (host logic only; torch's own ops, as in the reference) and -- marked `gpu` --
on the MI355X, where the aggregation goes through libdgll_hip.so.  Tolerance: 1e-4 (fp32), the bar BASELINE.json's
north_star states for layer activations; gradients of parameters are sums over many nodes and use 2e-3 relative."""
import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from dgll_amd import nn as dnn

DEVICES = [pytest.param("cpu", id="cpu"), pytest.param("cuda", id="gpu", marks=pytest.mark.gpu)]
ACT = dict(rtol=1e-4, atol=1e-5)
GRAD = dict(rtol=2e-3, atol=2e-4)


def close(actual, desired, **tol):
    np.testing.assert_allclose(actual.detach().float().cpu().numpy(), desired, **tol)


def sparse_adj(g, device):
    n = g.meta["n"]
    ind = torch.from_numpy(np.stack([g["adj_row"], g["adj_col"]]))
    return torch.sparse_coo_tensor(ind, g.t("adj_val"), (n, n)).to(device)


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("name", golden_names("gcn_conv_"))
def test_gcn_conv(name, device):
    g = load_golden(name)
    layer = dnn.gcnConv(g.meta["F"], g.meta["H"], bias="bias" in g)
    with torch.no_grad():
        layer.weight.copy_(g.t("weight"))
        if "bias" in g:
            layer.bias.copy_(g.t("bias"))
    layer = layer.to(device)
    adj = sparse_adj(g, device)
    x = g.t("x", device).requires_grad_()
    y = layer(x, adj)
    close(y, g["y"], **ACT)
    (y * g.t("gout", device)).sum().backward()
    close(x.grad, g["grad_x"], **GRAD)
    close(layer.weight.grad, g["grad_weight"], **GRAD)
    if "bias" in g:
        close(layer.bias.grad, g["grad_bias"], **GRAD)


@pytest.mark.parametrize("device", DEVICES)
def test_gcn_model(device):
    g = load_golden("gcn_model_s9")
    m = g.meta
    model = dnn.GCN(m["F"], m["nhid"], m["nclass"], dropout=0.5)
    model.load_state_dict({"gcn1.weight": g.t("w1"), "gcn1.bias": g.t("b1"), "gcn2.weight": g.t("w2"), "gcn2.bias": g.t("b2")})
    model = model.to(device).eval()
    x = g.t("x", device).requires_grad_()
    y = model(x, sparse_adj(g, device))
    close(y, g["y"], **ACT)
    (y * g.t("gout", device)).sum().backward()
    close(x.grad, g["grad_x"], **GRAD)
    close(model.gcn1.weight.grad, g["grad_w1"], **GRAD)
    close(model.gcn2.bias.grad, g["grad_b2"], **GRAD)


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("name", golden_names("sage_conv_"))
def test_sage_conv(name, device):
    g = load_golden(name)
    m = g.meta
    layer = dnn.sageConv(m["D"], m["H"], aggr_neighbor_method=m["aggr"], aggr_hid_method=m["hid"])
    layer.load_state_dict({"weight": g.t("weight"), "neighborAgg.weight": g.t("nbr_weight")})
    layer = layer.to(device)
    src = g.t("src", device).requires_grad_()
    nbr = g.t("nbr", device).requires_grad_()
    y = layer(src, nbr)
    close(y, g["y"], **ACT)
    (y * g.t("gout", device)).sum().backward()
    close(src.grad, g["grad_src"], **GRAD)
    close(nbr.grad, g["grad_nbr"], **GRAD)
    close(layer.weight.grad, g["grad_weight"], **GRAD)
    close(layer.neighborAgg.weight.grad, g["grad_nbr_weight"], **GRAD)


@pytest.mark.parametrize("device", DEVICES)
def test_sage_model(device):
    g = load_golden("sage_model_k3")
    m = g.meta
    model = dnn.GraphSage(m["D"], m["hidden"], m["num_neighbors"])
    model.load_state_dict({"gcn1.weight": g.t("w1"), "gcn1.neighborAgg.weight": g.t("wn1"),
                           "gcn2.weight": g.t("w2"), "gcn2.neighborAgg.weight": g.t("wn2")})
    model = model.to(device)
    y = model([g.t("h0", device), g.t("h1", device), g.t("h2", device)])
    close(y, g["y"], **ACT)


@pytest.mark.parametrize("device", DEVICES)
def test_sage_model_gradients(device):
    """a5, model level, backward: gradients of all three input hops and all four weights vs the imported reference."""
    g = load_golden("sage_model_k3_grads")
    m = g.meta
    model = dnn.GraphSage(m["D"], m["hidden"], m["num_neighbors"])
    model.load_state_dict({"gcn1.weight": g.t("w1"), "gcn1.neighborAgg.weight": g.t("wn1"),
                           "gcn2.weight": g.t("w2"), "gcn2.neighborAgg.weight": g.t("wn2")})
    model = model.to(device)
    hs = [g.t(k, device).requires_grad_() for k in ("h0", "h1", "h2")]
    y = model(hs)
    close(y, g["y"], **ACT)
    close(y, load_golden("sage_model_k3")["y"], **ACT)                 # the same tensors as the forward-only fixture
    (y * g.t("gout", device)).sum().backward()
    for h, k in zip(hs, ("grad_h0", "grad_h1", "grad_h2")):
        close(h.grad, g[k], **GRAD)
    for p, k in ((model.gcn1.weight, "grad_w1"), (model.gcn1.neighborAgg.weight, "grad_wn1"),
                 (model.gcn2.weight, "grad_w2"), (model.gcn2.neighborAgg.weight, "grad_wn2")):
        close(p.grad, g[k], **GRAD)


def test_sage_rejects_unknown_methods():
    with pytest.raises(ValueError):
        dnn.sageConv(4, 4, aggr_neighbor_method="median")(torch.zeros(2, 4), torch.zeros(2, 3, 4))
    with pytest.raises(ValueError):
        dnn.sageConv(4, 4, aggr_hid_method="mul")(torch.zeros(2, 4), torch.zeros(2, 3, 4))


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("name", golden_names("spgat_conv_") + golden_names("gat_conv_"))
def test_gat_layers(name, device):
    g = load_golden(name)
    m = g.meta
    cls = dnn.sparseGatConv if name.startswith("spgat") else dnn.gatConv
    layer = cls(m["Fin"], m["Fout"], dropout=0.0, alpha=m["alpha"], concat=m["concat"])
    layer.load_state_dict({"W": g.t("W"), "a": g.t("a")})
    layer = layer.to(device)
    x = g.t("x", device).requires_grad_()
    adj = g.t("adj", device, torch.float32)
    y = layer(x, adj)
    close(y, g["y"], **ACT)
    (y * g.t("gout", device)).sum().backward()
    close(x.grad, g["grad_x"], **GRAD)
    close(layer.W.grad, g["grad_W"], **GRAD)
    close(layer.a.grad, g["grad_a"], **GRAD)


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("kind", ["spgat", "gat"])
def test_multihead_models(kind, device):
    g = load_golden(kind + "_model_h8")
    m = g.meta
    cls = dnn.SpGAT if kind == "spgat" else dnn.GAT
    model = cls(m["nfeat"], m["nhid"], m["nclass"], dropout=0.6, alpha=m["alpha"], nheads=m["nheads"])
    sd = {"out_att.W": g.t("W_out"), "out_att.a": g.t("a_out")}
    for k in range(m["nheads"]):
        sd["attention_%d.W" % k] = g.t("W")[k]
        sd["attention_%d.a" % k] = g.t("a")[k]
    model.load_state_dict(sd)
    model = model.to(device).eval()
    x = g.t("x", device).requires_grad_()
    y = model(x, g.t("adj", device, torch.float32))
    close(y, g["y"], **ACT)
    (y * g.t("gout", device)).sum().backward()
    close(x.grad, g["grad_x"], **GRAD)
    close(torch.stack([a.W.grad for a in model.attentions]), g["grad_W"], **GRAD)
    close(torch.stack([a.a.grad for a in model.attentions]), g["grad_a"], **GRAD)
    close(model.out_att.W.grad, g["grad_W_out"], **GRAD)
    close(model.out_att.a.grad, g["grad_a_out"], **GRAD)


@pytest.mark.parametrize("device", DEVICES)
def test_special_spmm(device):
    g = load_golden("special_spmm_n120")
    n = g.meta["n"]
    edge = g.t("edge", device)
    values = g.t("values", device).requires_grad_()
    b = g.t("b", device).requires_grad_()
    y = dnn.SpecialSpmm()(edge, values, torch.Size([n, n]), b)
    close(y, g["y"], **ACT)
    (y * g.t("gout", device)).sum().backward()
    close(values.grad, g["grad_values"], **GRAD)
    close(b.grad, g["grad_b"], **GRAD)


def test_state_dict_names_match_reference():
    """Parameter names are the reference's (gcnconv.py:15-17, sageconv.py:20,61-63, gatconv.py:23-25,157-161)."""
    assert list(dnn.gcnConv(3, 4).state_dict()) == ["weight", "bias"]
    assert list(dnn.sageConv(3, 4).state_dict()) == ["weight", "neighborAgg.weight"]
    assert list(dnn.GraphSage(3, [4, 5], [2, 2]).state_dict()) == ["gcn1.weight", "gcn1.neighborAgg.weight",
                                                                 "gcn2.weight", "gcn2.neighborAgg.weight"]
    assert list(dnn.SpGAT(3, 4, 2, 0.5, 0.2, 2).state_dict()) == ["attention_0.W", "attention_0.a", "attention_1.W",
                                                                 "attention_1.a", "out_att.W", "out_att.a"]
    assert dnn.sparseGatConv(3, 4, 0.5, 0.2).a.shape == (1, 8) and dnn.gatConv(3, 4, 0.5, 0.2).a.shape == (8, 1)
