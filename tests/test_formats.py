"""On-disk formats in front of the layers and BASELINE config 1 on the bundled PPI data.

CPU: the loaders against fixtures produced by the reference's own loaders (tests/golden/gen_goldens.py gen_formats), and
the C oracle against the reference model's outputs on a real PPI graph.  GPU: the HIP-backed evaluation model against the
same goldens, forward and gradients."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import cref


def _write_citation(g, tmp_path):
    (tmp_path / "syn.content").write_text(str(g["content"]))
    (tmp_path / "syn.cites").write_text(str(g["cites"]))
    return str(tmp_path) + os.sep


def test_citation_loader_matches_reference_load_data(tmp_path):
    from dgll_amd.nn.utils import load_data

    g = load_golden("citation_format_n70")
    adj, features, labels, i_tr, i_va, i_te = load_data(_write_citation(g, tmp_path), "syn")
    adj = adj.coalesce()
    np.testing.assert_array_equal(adj.indices()[0].numpy(), g["adj_row"])
    np.testing.assert_array_equal(adj.indices()[1].numpy(), g["adj_col"])
    np.testing.assert_allclose(adj.values().numpy(), g["adj_val"], rtol=1e-6)
    np.testing.assert_allclose(features.numpy(), g["features"], rtol=1e-6)
    # class ids: same partition of the nodes (the reference's numbering follows set() order)
    mine, ref = labels.numpy(), g["labels"]
    assert len(set(zip(mine.tolist(), ref.tolist()))) == len(set(ref.tolist())) == len(set(mine.tolist()))
    names = g["class_names"]
    order = sorted(set(names.tolist()))
    np.testing.assert_array_equal(mine, [order.index(c) for c in names.tolist()])
    for a, b in ((i_tr, "idx_train"), (i_va, "idx_val"), (i_te, "idx_test")):
        np.testing.assert_array_equal(a.numpy(), g[b])


def test_citation_loader_rejects_unknown_paper(tmp_path):
    from dgll_amd.data.formats import load_citation

    g = load_golden("citation_format_n70")
    path = _write_citation(g, tmp_path)
    with open(path + "syn.cites", "a") as f:
        f.write("1000\t999999\n")
    with pytest.raises(ValueError):
        load_citation(path, "syn")


def test_sage_format_loader_matches_reference_loader(tmp_path):
    from dgll_amd.evaluation.ppi import load_ppi_dataset

    g = load_golden("sage_format_3graphs")
    (tmp_path / "valid_graph.json").write_text(str(g["graph_json"]))
    np.save(tmp_path / "valid_feats.npy", g["feats"])
    np.save(tmp_path / "valid_labels.npy", g["labels"])
    np.save(tmp_path / "valid_graph_id.npy", g["graph_id"])
    graphs = load_ppi_dataset(str(tmp_path), "valid")
    assert len(graphs) == g.meta["n_graphs"]
    for k, (ei, x, y) in enumerate(graphs):
        n = x.shape[0]
        assert ei.dtype == torch.int64 and x.dtype == torch.float32 and y.dtype == torch.float32
        keys = ei[0].numpy() * n + ei[1].numpy()
        np.testing.assert_array_equal(keys, g["g%d_edges" % k])          # sorted, unique, no self-loops
        np.testing.assert_array_equal(x.numpy(), g["g%d_x" % k])
        np.testing.assert_array_equal(y.numpy(), g["g%d_y" % k])


def _real(g):
    ei = g["edge_index"].astype(np.int64)
    return ei, g["x"], g["labels"].astype(np.float32), np.arange(0, g.meta["n"], 4)


def test_real_ppi_graph_c_oracle():
    """Evaluation/PPI/gcn_model.py on a graph of the bundled PPI test split, restated with the C oracle."""
    g = load_golden("ppi_real_test1_gcn2")
    ei, x, y, rows = _real(g)
    n = g.meta["n"]
    rowptr, col, val = cref.coo_to_csr(ei[0], ei[1], None, n)
    h = x
    for k in ("w0", "w1"):
        h = np.maximum(cref.spmm_csr(rowptr, col, val, cref.gemm(h, g[k])), 0)
    out = cref.gemm(h, g["w_out"].T.copy(), g["b_out"])
    scale = float(np.abs(g["out_rows"]).max())
    np.testing.assert_allclose(out[rows], g["out_rows"], rtol=1e-4, atol=1e-5 * scale)
    np.testing.assert_allclose(out.astype(np.float64).sum(0), g["out_colsum"], rtol=1e-4, atol=1e-4 * scale)
    t = torch.from_numpy(out)
    loss = torch.nn.CrossEntropyLoss()(t, torch.from_numpy(y))
    np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ppi_gcn_3layer", "ppi_real_test1_gcn2"])
def test_ppi_evaluation_model_on_gpu(name, cuda_device):
    from dgll_amd.evaluation.ppi import GCN

    g = load_golden(name)
    real = name.startswith("ppi_real")
    layers = 2 if real else 3
    ei = torch.from_numpy(g["edge_index"].astype(np.int64)).to(cuda_device)
    x = g.t("x").to(cuda_device).requires_grad_(True)
    model = GCN(x.shape[1], g["w0"].shape[1], g["w_out"].shape[0], num_layers=layers).to(cuda_device)
    state = {"layers.%d.weight" % i: g.t("w%d" % i) for i in range(layers)}
    state.update({"out_layer.weight": g.t("w_out"), "out_layer.bias": g.t("b_out")})
    model.load_state_dict(state)                   # the reference's parameter names
    out = model(ei, x)
    params = [x] + [l.weight for l in model.layers] + [model.out_layer.weight, model.out_layer.bias]
    if real:
        rows = torch.arange(0, g.meta["n"], 4, device=cuda_device)
        scale = float(np.abs(g["out_rows"]).max())
        np.testing.assert_allclose(out[rows].detach().cpu().numpy(), g["out_rows"], rtol=1e-4, atol=1e-5 * scale)
        loss = torch.nn.CrossEntropyLoss()(out, torch.from_numpy(g["labels"].astype(np.float32)).to(cuda_device))
        np.testing.assert_allclose(float(loss.detach()), float(g["loss"]), rtol=1e-4)
        grads = torch.autograd.grad(loss, params)
        expect = [g["grad_x_rows"], g["grad_w0"], g["grad_w1"], g["grad_w_out"], g["grad_b_out"]]
        grads = [grads[0][rows]] + list(grads[1:])
    else:
        scale = float(np.abs(g["y"]).max())
        np.testing.assert_allclose(out.detach().cpu().numpy(), g["y"], rtol=1e-4, atol=1e-5 * scale)
        grads = torch.autograd.grad((out * g.t("gout").to(cuda_device)).sum(), params)
        expect = [g["grad_x"], g["grad_w0"], g["grad_w1"], g["grad_w2"], g["grad_w_out"], g["grad_b_out"]]
    for got, ref in zip(grads, expect):
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-4, atol=2e-5 * float(np.abs(ref).max()))
    out2 = model(ei, x)                            # second call: the cached CSR, same result bit for bit
    assert torch.equal(out2, out)


def _toy_dataset(n=60, e=400, f=7, c=5, seed=0):
    rng = np.random.default_rng(seed)
    src, dst = rng.integers(0, n, e), rng.integers(0, n, e)
    feat = rng.standard_normal((n, f)).astype(np.float32)
    label = rng.integers(0, c, n)
    return src, dst, feat, label


def _expect_adj(src, dst, n, symmetrise):
    s = set(zip(src.tolist(), dst.tolist()))
    if symmetrise:
        s |= set(zip(dst.tolist(), src.tolist()))
    return [sorted(u for (v2, u) in s if v2 == v) for v in range(n)]


@pytest.mark.parametrize("layout", ["edge_index", "src_dst", "csr"])
@pytest.mark.parametrize("symmetrise", [False, True])
def test_edge_list_npz_loader(tmp_path, layout, symmetrise):
    """Raw edge-list datasets (ogbn-products' processed dict, PyG-style dumps) -> DGraph, for every key spelling."""
    from dgll_amd.data import formats

    src, dst, feat, label = _toy_dataset()
    n = feat.shape[0]
    arrays = dict(node_feat=feat, node_label=label.reshape(-1, 1), train_idx=np.arange(0, 30), valid_idx=np.arange(30, 45),
                  test_idx=np.arange(45, 60))
    if layout == "edge_index":
        arrays["edge_index"] = np.stack([src, dst])
    elif layout == "src_dst":
        arrays.update(src=src, dst=dst)
        arrays["x"] = arrays.pop("node_feat")
        arrays["y"] = arrays.pop("node_label")[:, 0]
    else:
        import scipy.sparse as sp

        a = sp.csr_matrix((np.ones(src.size), (src, dst)), shape=(n, n))
        a.sum_duplicates()
        a.sort_indices()
        arrays.update(indptr=a.indptr, indices=a.indices)
    path = tmp_path / "toy.npz"
    np.savez(path, **arrays)
    g = formats.load_node_dataset(str(path), symmetrise=symmetrise)
    want = _expect_adj(src, dst, n, symmetrise)
    assert [g.edges[v] for v in range(n)] == want
    assert g.get_neighbors(torch.tensor([3, 7])) == [want[3], want[7]]
    np.testing.assert_array_equal(g.features.numpy(), feat)
    np.testing.assert_array_equal(g.labels.numpy(), label)
    assert g.get_train_nodes().tolist() == list(range(30)) and g.get_test_nodes().tolist() == list(range(45, 60))
    csr = g.to_csr("cpu")
    assert csr.nnz == sum(len(w) for w in want) and csr.n_rows == n


def test_reddit_and_ogb_raw_loaders(tmp_path):
    import gzip

    from dgll_amd.data import formats

    src, dst, feat, label = _toy_dataset(n=40, e=300, seed=3)
    n = feat.shape[0]
    # DGL's RedditDataset files
    rd = tmp_path / "reddit"
    rd.mkdir()
    types = np.array([1] * 25 + [2] * 5 + [3] * 10)
    np.savez(rd / "reddit_data.npz", feature=feat, label=label, node_types=types)
    np.savez(rd / "reddit_graph.npz", row=src, col=dst, data=np.ones(src.size), shape=np.array([n, n]))
    g = formats.load_node_dataset(str(rd))
    assert [g.edges[v] for v in range(n)] == _expect_adj(src, dst, n, False)
    assert g.get_train_nodes().tolist() == list(range(25)) and g.get_validation_nodes().tolist() == list(range(25, 30))
    # OGB raw csv files (gzipped), undirected-once edge list
    od = tmp_path / "ogbn_toy"
    (od / "raw").mkdir(parents=True)
    (od / "split" / "sales_ranking").mkdir(parents=True)
    for stem, arr, fmt in (("edge", np.stack([src, dst], 1), "%d"), ("node-feat", feat, "%.8g"), ("node-label", label.reshape(-1, 1), "%d")):
        with gzip.open(od / "raw" / (stem + ".csv.gz"), "wt") as f:
            np.savetxt(f, arr, delimiter=",", fmt=fmt)
    for stem, idx in (("train", np.arange(0, 20)), ("valid", np.arange(20, 30)), ("test", np.arange(30, 40))):
        with gzip.open(od / "split" / "sales_ranking" / (stem + ".csv.gz"), "wt") as f:
            np.savetxt(f, idx, fmt="%d")
    g2 = formats.load_node_dataset(str(od))
    assert [g2.edges[v] for v in range(n)] == _expect_adj(src, dst, n, True)
    np.testing.assert_allclose(g2.features.numpy(), feat, rtol=1e-6)
    np.testing.assert_array_equal(g2.labels.numpy(), label)
    assert g2.get_test_nodes().tolist() == list(range(30, 40))
