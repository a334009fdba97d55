#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only where /root/reference exists (the build container); the GPU box only sees the .npz files
this script wrote.  Recipe:

  1. ``import dgll`` from /root/reference, then replace ``dgll.backend`` (which is literally ``torch``,
     dgll/__init__.py:1) by a module that forwards to torch and adds the six names the layers use but
     torch lacks: Parameter, init, LeakyReLU, Dropout, elu, dropout.
  2. Load dgll/nn/Convolution/{gcnconv,sageconv,gatconv}.py, dgll/nn/utils/utils.py,
     dgll/data/dgraph.py, dgll/sampling/*.py and Evaluation/PPI/gcn_model.py BY FILE PATH (the package
     __init__ files are broken, Convolution/__init__.py:1-5).
  3. sageconv.py:33-38 discards its reduction result; the documented one-line fix ("assign the
     result") is applied in memory by replacing NeighborAggregator.forward, and sageConv.weight (left
     uninitialised, sageconv.py:63,67-68) is initialised by calling reset_parameters().
  4. Run each layer on small seeded inputs and store inputs, parameters, outputs and gradients.

Only data is written: no reference source text is stored in the fixtures.

Usage:  python tests/golden/gen_goldens.py            (writes next to this file)
"""
import contextlib
import importlib.util
import io
import json
import os
import random
import sys
import types

import numpy as np
import scipy.sparse as sp
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

SAGE_FIX = ("NeighborAggregator.forward: assign the reduction result "
            "(mean(dim=1) / sum(dim=1) / max(dim=1)[0]) before F.matmul; "
            "sageConv.reset_parameters() called after construction")


# ---------------------------------------------------------------------------------- reference import
def load_reference():
    assert os.path.isdir(REF), "reference not mounted; goldens can only be regenerated in the build container"
    sys.path.insert(0, REF)
    for k in [k for k in sys.modules if k == "dgll" or k.startswith("dgll.")]:
        del sys.modules[k]
    import dgll  # the reference package: `import torch as backend`

    assert dgll.__file__.startswith(REF), dgll.__file__

    shim = types.ModuleType("dgll.backend")
    shim.__getattr__ = lambda name: getattr(torch, name)
    shim.Parameter = torch.nn.Parameter
    shim.init = torch.nn.init
    shim.LeakyReLU = torch.nn.LeakyReLU
    shim.Dropout = torch.nn.Dropout
    shim.elu = torch.nn.functional.elu
    shim.dropout = torch.nn.functional.dropout
    dgll.backend = shim
    sys.modules["dgll.backend"] = shim

    def by_path(name, rel, package=None):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        if package:
            mod.__package__ = package
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    ref = types.SimpleNamespace()
    ref.gcnconv = by_path("ref_gcnconv", "dgll/nn/Convolution/gcnconv.py")
    ref.sageconv = by_path("ref_sageconv", "dgll/nn/Convolution/sageconv.py")
    ref.gatconv = by_path("ref_gatconv", "dgll/nn/Convolution/gatconv.py")
    ref.utils = by_path("ref_nn_utils", "dgll/nn/utils/utils.py")
    ref.ppi_model = by_path("ref_ppi_gcn_model", "Evaluation/PPI/gcn_model.py")
    ref.ppi_loader = by_path("ref_ppi_dataloader", "Evaluation/PPI/ppi_dataloader.py")
    # dgll.data / dgll.sampling use relative imports of the backend: give them a package home.
    pkg_data = types.ModuleType("dgll.data")
    pkg_data.__path__ = [os.path.join(REF, "dgll/data")]
    sys.modules["dgll.data"] = pkg_data
    ref.dgraph = by_path("dgll.data.dgraph", "dgll/data/dgraph.py", package="dgll.data")
    pkg_s = types.ModuleType("dgll.sampling")
    pkg_s.__path__ = [os.path.join(REF, "dgll/sampling")]
    sys.modules["dgll.sampling"] = pkg_s
    ref.base_sampler = by_path("dgll.sampling.base_sampler", "dgll/sampling/base_sampler.py", package="dgll.sampling")
    ref.dgllsampler = by_path("dgll.sampling.dgllsampler", "dgll/sampling/dgllsampler.py", package="dgll.sampling")

    # --- the documented SAGE fix (in memory only) ---
    F = shim

    def fixed_forward(self, neighbor_feature):
        if self.aggr_method == "mean":
            neighbor_feature = neighbor_feature.mean(dim=1)
        elif self.aggr_method == "sum":
            neighbor_feature = neighbor_feature.sum(dim=1)
        elif self.aggr_method == "max":
            neighbor_feature = neighbor_feature.max(dim=1)[0]
        else:
            raise ValueError("Unsupported aggr_method, expected mean, sum, max, but got {}".format(self.aggr_method))
        neighbor_hidden = F.matmul(neighbor_feature, self.weight)
        if self.use_bias:
            neighbor_hidden += self.bias
        return neighbor_hidden

    ref.sageconv.NeighborAggregator.forward = fixed_forward
    return ref


# ---------------------------------------------------------------------------------- inputs
def rmat_edges(scale, edge_factor, seed, a=0.57, b=0.19, c=0.19):
    """Bit-per-level RMAT, numpy default_rng(seed); directed, with duplicates."""
    rng = np.random.default_rng(seed)
    n_edges = edge_factor << scale
    src = np.zeros(n_edges, dtype=np.int64)
    dst = np.zeros(n_edges, dtype=np.int64)
    for _ in range(scale):
        r = rng.random(n_edges)
        src = (src << 1) | (r >= a + b)
        dst = (dst << 1) | (((r >= a) & (r < a + b)) | (r >= a + b + c))
    return src, dst


def ref_normalised_adj(ref, src, dst, n, self_loops=True):
    """adjacency prep exactly as nn/utils/utils.py:163-171,240-257 does it (a11)."""
    adj = sp.coo_matrix((np.ones(src.shape[0]), (src, dst)), shape=(n, n), dtype=np.float32)
    adj.sum_duplicates()
    adj.data[:] = 1.0
    adj = adj + adj.T.multiply(adj.T > adj) - adj.multiply(adj.T > adj)  # utils.py:164
    if self_loops:
        adj = adj + sp.eye(adj.shape[0])
    adj = ref.utils.normalize(adj)  # utils.py:171 / :240-247
    return ref.utils.sparse_mx_to_torch_sparse_tensor(adj)  # utils.py:250-257


def save(name, meta, **arrays):
    meta = dict(meta)
    meta.update(torch=torch.__version__, numpy=np.__version__, python=sys.version.split()[0])
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %-34s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def grads_of(out, gout, tensors):
    loss = (out * gout).sum()
    gs = torch.autograd.grad(loss, tensors, allow_unused=True)
    return [torch.zeros_like(t) if g is None else g for g, t in zip(gs, tensors)]


# ---------------------------------------------------------------------------------- cases
def gen_gcn(ref):
    cases = [
        # name, scale, ef, F, H, bias, self_loops/normalised
        ("gcn_conv_s8_f7_h16", 8, 8, 7, 16, True, True),
        ("gcn_conv_s10_f64_h47", 10, 16, 64, 47, True, True),
        ("gcn_conv_s9_f33_h16_raw", 9, 4, 33, 16, False, False),
        ("gcn_conv_s9_f128_h256", 9, 16, 128, 256, True, True),
    ]
    for name, scale, ef, Fi, H, bias, norm in cases:
        torch.manual_seed(1)
        n = 1 << scale
        src, dst = rmat_edges(scale, ef, seed=scale)
        if norm:
            adj = ref_normalised_adj(ref, src, dst, n)
        else:
            # raw power-law adjacency, random weights, rows with no entries stay empty (no self-loops)
            m = sp.coo_matrix((np.ones(src.shape[0], np.float32), (src, dst)), shape=(n, n))
            m.sum_duplicates()
            m = m.tocsr().tocoo()
            rng = np.random.default_rng(7)
            vals = rng.standard_normal(m.nnz).astype(np.float32)
            adj = torch.sparse_coo_tensor(np.vstack((m.row, m.col)).astype(np.int64), vals, (n, n))
        layer = ref.gcnconv.gcnConv(Fi, H, bias=bias)
        x = torch.randn(n, Fi, requires_grad=True)
        y = layer(x, adj)
        gout = torch.randn_like(y)
        params = [x, layer.weight] + ([layer.bias] if bias else [])
        gs = grads_of(y, gout, params)
        ind = adj._indices()
        arrays = dict(adj_row=ind[0], adj_col=ind[1], adj_val=adj._values(), x=x, weight=layer.weight,
                      y=y, gout=gout, grad_x=gs[0], grad_weight=gs[1])
        if bias:
            arrays.update(bias=layer.bias, grad_bias=gs[2])
        deg = np.bincount(ind[0].numpy(), minlength=n)
        save(name, dict(row="a1", ref="dgll/nn/Convolution/gcnconv.py:29-35", n=n, F=Fi, H=H, nnz=int(ind.shape[1]),
                        zero_degree_rows=int((deg == 0).sum()), max_degree=int(deg.max()), normalised=norm,
                        adj_prep="nn/utils/utils.py:164,171,240-257" if norm else "raw"), **arrays)

    # a2: the 2-layer GCN model, eval mode (dropout inactive) and its log_softmax output
    torch.manual_seed(2)
    scale, n, Fi, nhid, ncls = 9, 512, 50, 64, 7
    src, dst = rmat_edges(scale, 8, seed=21)
    adj = ref_normalised_adj(ref, src, dst, n)
    model = ref.gcnconv.GCN(Fi, nhid, ncls, dropout=0.5).eval()
    x = torch.randn(n, Fi, requires_grad=True)
    y = model(x, adj)
    gout = torch.randn_like(y)
    ps = [x, model.gcn1.weight, model.gcn1.bias, model.gcn2.weight, model.gcn2.bias]
    gs = grads_of(y, gout, ps)
    ind = adj._indices()
    save("gcn_model_s9", dict(row="a2", ref="dgll/nn/Convolution/gcnconv.py:43-58", n=n, F=Fi, nhid=nhid, nclass=ncls,
                              mode="eval"),
         adj_row=ind[0], adj_col=ind[1], adj_val=adj._values(), x=x, y=y, gout=gout,
         w1=ps[1], b1=ps[2], w2=ps[3], b2=ps[4], grad_x=gs[0], grad_w1=gs[1], grad_b1=gs[2], grad_w2=gs[3],
         grad_b2=gs[4])


def gen_sage(ref):
    cases = [
        ("sage_conv_mean_sum", 50, 10, 7, 16, "mean", "sum"),
        ("sage_conv_mean_concat", 33, 25, 64, 47, "mean", "concat"),
        ("sage_conv_sum_k1", 20, 1, 5, 8, "sum", "sum"),
        ("sage_conv_max_sum", 40, 10, 16, 16, "max", "sum"),
        ("sage_conv_mean_f100_h256", 64, 10, 100, 256, "mean", "sum"),
    ]
    for name, N, K, D, H, aggr, hid in cases:
        torch.manual_seed(3)
        layer = ref.sageconv.sageConv(D, H, aggr_neighbor_method=aggr, aggr_hid_method=hid)
        layer.reset_parameters()
        src = torch.randn(N, D, requires_grad=True)
        nbr = torch.randn(N, K, D, requires_grad=True)
        y = layer(src, nbr)
        gout = torch.randn_like(y)
        ps = [src, nbr, layer.weight, layer.neighborAgg.weight]
        gs = grads_of(y, gout, ps)
        save(name, dict(row="a3+a4", ref="dgll/nn/Convolution/sageconv.py:32-45,70-83", N=N, K=K, D=D, H=H,
                        aggr=aggr, hid=hid, fix=SAGE_FIX),
             src=src, nbr=nbr, weight=layer.weight, nbr_weight=layer.neighborAgg.weight, y=y, gout=gout,
             grad_src=gs[0], grad_nbr=gs[1], grad_weight=gs[2], grad_nbr_weight=gs[3])

    # a5: the 2-layer GraphSage hop pyramid (equal fan-outs are the only shapes the reference accepts)
    torch.manual_seed(4)
    D, hidden, K, n0 = 12, [16, 8], [3, 3], 10
    model = ref.sageconv.GraphSage(D, hidden, K)
    model.gcn1.reset_parameters()
    model.gcn2.reset_parameters()
    feats = [torch.randn(n0, D), torch.randn(n0 * 3, D), torch.randn(n0 * 9, D)]
    y = model(feats)
    save("sage_model_k3", dict(row="a5", ref="dgll/nn/Convolution/sageconv.py:86-114", D=D, hidden=hidden,
                               num_neighbors=K, fix=SAGE_FIX),
         h0=feats[0], h1=feats[1], h2=feats[2], y=y,
         w1=model.gcn1.weight, wn1=model.gcn1.neighborAgg.weight,
         w2=model.gcn2.weight, wn2=model.gcn2.neighborAgg.weight)


def dense_adj(n, density, seed, self_loops=True):
    rng = np.random.default_rng(seed)
    a = (rng.random((n, n)) < density).astype(np.float32)
    if self_loops:
        np.fill_diagonal(a, 1.0)
    return torch.from_numpy(a)


def gen_gat(ref):
    # a6: sparseGatConv
    for name, n, Fi, Fo, concat, dens in [
        ("spgat_conv_n200_f19_o32", 200, 19, 32, True, 0.05),
        ("spgat_conv_n150_f64_o8_noconcat", 150, 64, 8, False, 0.1),
        ("spgat_conv_n97_f5_o3", 97, 5, 3, True, 0.3),
    ]:
        torch.manual_seed(5)
        adj = dense_adj(n, dens, seed=n)
        layer = ref.gatconv.sparseGatConv(Fi, Fo, dropout=0.0, alpha=0.2, concat=concat)
        x = torch.randn(n, Fi, requires_grad=True)
        y = layer(x, adj)
        gout = torch.randn_like(y)
        gs = grads_of(y, gout, [x, layer.W, layer.a])
        save(name, dict(row="a6", ref="dgll/nn/Convolution/gatconv.py:89-151", n=n, Fin=Fi, Fout=Fo, alpha=0.2,
                        concat=concat, dropout=0.0),
             adj=adj.to(torch.uint8), x=x, W=layer.W, a=layer.a, y=y, gout=gout,
             grad_x=gs[0], grad_W=gs[1], grad_a=gs[2])

    # a7: SpecialSpmmFunction forward/backward
    torch.manual_seed(6)
    n, Fo = 120, 24
    adj = dense_adj(n, 0.08, seed=77)
    edge = adj.nonzero().t()
    values = torch.rand(edge.shape[1], requires_grad=True)
    b = torch.randn(n, Fo, requires_grad=True)
    y = ref.gatconv.SpecialSpmm()(edge, values, torch.Size([n, n]), b)
    gout = torch.randn_like(y)
    gs = grads_of(y, gout, [values, b])
    save("special_spmm_n120", dict(row="a7", ref="dgll/nn/Convolution/gatconv.py:60-86", n=n, Fout=Fo),
         edge=edge, values=values, b=b, y=y, gout=gout, grad_values=gs[0], grad_b=gs[1])

    # a8: dense gatConv
    for name, n, Fi, Fo, concat in [("gat_conv_n200_f19_o32", 200, 19, 32, True),
                                    ("gat_conv_n64_f8_o7_noconcat", 64, 8, 7, False)]:
        torch.manual_seed(7)
        adj = dense_adj(n, 0.05, seed=n + 1)
        layer = ref.gatconv.gatConv(Fi, Fo, dropout=0.0, alpha=0.2, concat=concat)
        x = torch.randn(n, Fi, requires_grad=True)
        y = layer(x, adj)
        gout = torch.randn_like(y)
        gs = grads_of(y, gout, [x, layer.W, layer.a])
        save(name, dict(row="a8", ref="dgll/nn/Convolution/gatconv.py:10-57", n=n, Fin=Fi, Fout=Fo, alpha=0.2,
                        concat=concat, dropout=0.0),
             adj=adj.to(torch.uint8), x=x, W=layer.W, a=layer.a, y=y, gout=gout,
             grad_x=gs[0], grad_W=gs[1], grad_a=gs[2])

    # a9: 8-head models in eval mode
    for kind, cls in [("spgat", ref.gatconv.SpGAT), ("gat", ref.gatconv.GAT)]:
        torch.manual_seed(8)
        n, nfeat, nhid, ncls, heads = 160, 19, 8, 7, 8
        adj = dense_adj(n, 0.06, seed=99)
        model = cls(nfeat, nhid, ncls, dropout=0.6, alpha=0.2, nheads=heads).eval()
        x = torch.randn(n, nfeat, requires_grad=True)
        y = model(x, adj)
        gout = torch.randn_like(y)
        ps = [x] + [p for att in model.attentions for p in (att.W, att.a)] + [model.out_att.W, model.out_att.a]
        gs = grads_of(y, gout, ps)
        arrays = dict(adj=adj.to(torch.uint8), x=x, y=y, gout=gout, grad_x=gs[0],
                      W=torch.stack([att.W for att in model.attentions]),
                      a=torch.stack([att.a for att in model.attentions]),
                      grad_W=torch.stack(gs[1:1 + 2 * heads:2]), grad_a=torch.stack(gs[2:2 + 2 * heads:2]),
                      W_out=model.out_att.W, a_out=model.out_att.a, grad_W_out=gs[-2], grad_a_out=gs[-1])
        save(kind + "_model_h8", dict(row="a9", ref="dgll/nn/Convolution/gatconv.py:154-199", n=n, nfeat=nfeat,
                                      nhid=nhid, nclass=ncls, nheads=heads, alpha=0.2, mode="eval"), **arrays)


def gen_ppi(ref):
    """Config 1 plumbing: Evaluation/PPI/gcn_model.py GCN on a PPI-shaped graph (duplicated directed edges,
    all-ones uncoalesced COO rebuilt per layer, gcn_model.py:56,73)."""
    torch.manual_seed(9)
    n, Fi, Hd, ncls = 300, 50, 64, 121
    src, dst = rmat_edges(8, 12, seed=5)
    src, dst = src % n, dst % n
    keep = src != dst
    edge_index = torch.from_numpy(np.vstack((src[keep], dst[keep])))
    model = ref.ppi_model.GCN(Fi, Hd, ncls, num_layers=3)
    with torch.no_grad():  # the reference's randn init gives activations ~1e4; scale to keep fp32 comparisons meaningful
        for layer in model.layers:
            layer.weight.mul_(0.1)
    x = torch.randn(n, Fi, requires_grad=True)
    y = model(edge_index, x)
    gout = torch.randn_like(y)
    ps = [x] + [l.weight for l in model.layers] + [model.out_layer.weight, model.out_layer.bias]
    gs = grads_of(y, gout, ps)
    save("ppi_gcn_3layer", dict(row="a1 (config 1)", ref="Evaluation/PPI/gcn_model.py:63-94", n=n, F=Fi, hidden=Hd,
                                nclass=ncls, duplicate_edges=True),
         edge_index=edge_index, x=x, y=y, gout=gout, w0=ps[1], w1=ps[2], w2=ps[3], w_out=ps[4], b_out=ps[5],
         grad_x=gs[0], grad_w0=gs[1], grad_w1=gs[2], grad_w2=gs[3], grad_w_out=gs[4], grad_b_out=gs[5])


def gen_sampler(ref):
    """Neighbour sampler IDs under random.seed(s): dgll/sampling/base_sampler.py:30-58, dgllsampler.py:10-21."""
    rng = np.random.default_rng(11)
    n = 400
    edges = []
    for v in range(n):
        deg = int(min(n - 1, rng.zipf(1.6))) if v % 17 else 0  # power-law degrees, some isolated nodes
        edges.append(sorted(rng.choice(n, size=deg, replace=False).tolist()))
    g = ref.dgraph.DGraph(nodes=torch.arange(n), edges=edges, labels=torch.arange(n) % 7,
                          features=torch.arange(n * 3, dtype=torch.float32).view(n, 3))
    flat = np.concatenate([np.asarray(e, dtype=np.int64) for e in edges] + [np.zeros(0, np.int64)])
    ptr = np.cumsum([0] + [len(e) for e in edges])
    arrays = dict(adj_ptr=ptr, adj_idx=flat)
    metas = []
    for ci, (seed, fanouts, seeds) in enumerate([
        (0, [3, 2], [0, 1, 2, 5, 17, 34, 399]),
        (1, [10, 25], list(range(100, 164))),
        (2, [25, 10, 10], [7, 8, 9, 10, 11]),
        (3, [1], [3, 3, 3]),
        (4, [], [1, 2, 3]),
    ]):
        sampler = ref.dgllsampler.DGLLNeighborSampler(fanouts)
        random.seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):  # dgllsampler.py:13 has a stray print
            if fanouts:
                inp, outp, subgs = sampler.sample(g, torch.tensor(seeds))
            else:
                # empty fan-out list: the loop body never runs and `input_nodes` is unbound (dgllsampler.py:21)
                try:
                    sampler.sample(g, torch.tensor(seeds))
                    raised = False
                except UnboundLocalError:
                    raised = True
                metas.append(dict(seed=seed, fanouts=fanouts, seeds=seeds, raises_unbound_local=raised))
                continue
        arrays["c%d_input_nodes" % ci] = inp
        arrays["c%d_output_nodes" % ci] = outp
        for li, sg in enumerate(subgs):
            arrays["c%d_l%d_src" % (ci, li)] = sg.src_nodes()
            arrays["c%d_l%d_dst" % (ci, li)] = sg.dst_nodes()
            arrays["c%d_l%d_nodes" % (ci, li)] = sg.nodes()
        metas.append(dict(seed=seed, fanouts=fanouts, seeds=seeds, n_layers=len(subgs)))
    # DGraph queries (example.py:22-41)
    q = torch.tensor([0, 2, 5, 6, 9, 23])
    arrays["q_nodes"] = q
    arrays["q_induced"] = g.get_induced_subgraph(q)
    arrays["q_features"] = g.get_features(q)
    arrays["q_labels"] = g.get_labels(q)
    save("sampler_n400", dict(row="f1", ref="dgll/sampling/base_sampler.py:30-58; dgllsampler.py:10-21; "
                                            "data/dgraph.py:49-105", n=n, cases=metas), **arrays)


def gen_adj_prep(ref):
    """a11: D^-1 (A + I) and the COO hand-over, nn/utils/utils.py:164,171,240-257."""
    src, dst = rmat_edges(7, 6, seed=3)
    n = 128
    adj = ref_normalised_adj(ref, src, dst, n)
    ind = adj._indices()
    save("adj_prep_s7", dict(row="a11", ref="dgll/nn/utils/utils.py:164,171,240-257", n=n,
                             is_coalesced=bool(adj.is_coalesced())),
         src=src, dst=dst, adj_row=ind[0], adj_col=ind[1], adj_val=adj._values())


def _edge_set(edge_index, n):
    e = edge_index.numpy() if isinstance(edge_index, torch.Tensor) else edge_index
    return np.unique(e[0].astype(np.int64) * n + e[1])


def gen_formats(ref):
    """f4: the two on-disk formats in front of the layers, run through the reference's own loaders.
    (1) Cora-format text -> utils.py:146-185 load_data; (2) GraphSAGE-format PPI directory ->
    Evaluation/PPI/ppi_dataloader.py:10-61; (3) one graph of the REAL bundled PPI data (Evaluation/PPI.tar.xz) with the
    reference's Evaluation/PPI/gcn_model.py run on it -- BASELINE config 1 on its own data."""
    import tarfile
    import tempfile
    import warnings

    warnings.simplefilter("ignore")
    rng = np.random.default_rng(21)
    with tempfile.TemporaryDirectory() as tmp:
        # ---- (1) citation text: non-contiguous ids, a repeated citation, a mutual pair, a self citation, an empty row
        n, nf = 70, 23
        ids = rng.permutation(np.arange(1000, 1000 + 5 * n, 5))[:n]
        feats = (rng.random((n, nf)) < 0.2).astype(np.int64)
        feats[7] = 0
        names = np.array(["Theory", "Neural_Nets", "Case_Based", "Rule_Learning", "Genetic"])[rng.integers(0, 5, n)]
        pairs = ids[rng.integers(0, n, (260, 2))]
        pairs = np.concatenate([pairs, pairs[:3], pairs[5:8, ::-1], np.array([[ids[4], ids[4]]])])
        content = "\n".join("%d\t%s\t%s" % (ids[i], "\t".join(map(str, feats[i])), names[i]) for i in range(n)) + "\n"
        cites = "\n".join("%d\t%d" % (a, b) for a, b in pairs) + "\n"
        open(os.path.join(tmp, "syn.content"), "w").write(content)
        open(os.path.join(tmp, "syn.cites"), "w").write(cites)
        with contextlib.redirect_stdout(io.StringIO()):
            adj, features, labels, i_tr, i_va, i_te = ref.utils.load_data(path=tmp + "/", dataset="syn")
        adj = adj.coalesce()
        save("citation_format_n70", dict(row="f4", ref="dgll/nn/utils/utils.py:146-185,240-257", n=n, nfeat=nf,
                                         note="labels: the reference numbers classes in set() order (hash-seed "
                                              "dependent); compare as a partition, not by id"),
             content=np.array(content), cites=np.array(cites), adj_row=adj.indices()[0], adj_col=adj.indices()[1],
             adj_val=adj.values(), features=features, labels=labels, class_names=names, idx_train=i_tr, idx_val=i_va,
             idx_test=i_te)

        # ---- (2) GraphSAGE-format directory: 3 graphs, undirected node-link file with a self-loop and a repeated link
        sizes = [17, 9, 30]
        gid = np.repeat(np.array([4, 5, 6]), sizes)
        tot = int(sum(sizes))
        links, base = [], 0
        for sz in sizes:
            e = rng.integers(0, sz, (3 * sz, 2)) + base
            e[:, 0] = np.maximum(e[:, 0], base + 0)
            links += [dict(source=int(a), target=int(b)) for a, b in e]
            links.append(dict(source=base, target=base + 1))            # the first node always has an edge
            base += sz
        links.append(dict(source=3, target=3))
        links.append(dict(links[0]))
        node_link = dict(directed=False, multigraph=False, graph={}, nodes=[dict(id=i) for i in range(tot)], links=links)
        d = os.path.join(tmp, "sage")
        os.makedirs(d)
        gjson = json.dumps(node_link)
        open(os.path.join(d, "valid_graph.json"), "w").write(gjson)
        fx = rng.standard_normal((tot, 6))
        fy = (rng.random((tot, 4)) < 0.3).astype(np.int64)
        np.save(os.path.join(d, "valid_feats.npy"), fx)
        np.save(os.path.join(d, "valid_labels.npy"), fy)
        np.save(os.path.join(d, "valid_graph_id.npy"), gid)
        graphs = ref.ppi_loader.load_ppi_dataset(d, "valid")
        arrays = dict(graph_json=np.array(gjson), feats=fx, labels=fy, graph_id=gid)
        for k, (ei, x, y) in enumerate(graphs):
            arrays["g%d_edges" % k] = _edge_set(ei, x.shape[0])       # order-free: key = src * n + dst, sorted
            arrays["g%d_x" % k] = x
            arrays["g%d_y" % k] = y
        save("sage_format_3graphs", dict(row="f4", ref="Evaluation/PPI/ppi_dataloader.py:10-61", n_graphs=len(graphs),
                                         sizes=sizes, edges="sorted unique keys src*n+dst per graph"), **arrays)

        # ---- (3) the real PPI data: cross-check the build's loader on all three splits, keep one graph as a fixture
        with tarfile.open(os.path.join(REF, "Evaluation/PPI.tar.xz")) as tf:
            tf.extractall(tmp)
        real = os.path.join(tmp, "PPI")
        spec = importlib.util.spec_from_file_location("build_formats", os.path.join(os.path.dirname(OUT), "..", "dgll_amd", "data", "formats.py"))
        mine = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mine)
        stats = {}
        for split in ("train", "valid", "test"):
            a = ref.ppi_loader.load_ppi_dataset(real, split)
            b = mine.load_ppi_dataset(real, split)
            assert len(a) == len(b)
            for (e1, x1, y1), (e2, x2, y2) in zip(a, b):
                assert torch.equal(x1, x2) and torch.equal(y1, y2)
                assert np.array_equal(_edge_set(e1, x1.shape[0]), _edge_set(e2, x1.shape[0]))
                assert e1.shape == e2.shape
            stats[split] = dict(graphs=len(a), nodes=int(sum(x.shape[0] for _, x, _ in a)), edges=int(sum(e.shape[1] for e, _, _ in a)))
            if split == "test":
                ei, x, y = a[1]
        print("real PPI: build loader == reference loader on", stats)
        torch.manual_seed(13)
        model = ref.ppi_model.GCN(x.shape[1], 64, y.shape[1], num_layers=2)      # config 1: 2-layer GCN
        with torch.no_grad():
            for layer in model.layers:
                layer.weight.mul_(0.05)
        xg = x.clone().requires_grad_(True)
        out = model(ei, xg)
        loss = torch.nn.CrossEntropyLoss()(out, y)                                 # train_gcn.py:27,45
        ps = [xg] + [l.weight for l in model.layers] + [model.out_layer.weight, model.out_layer.bias]
        gs = torch.autograd.grad(loss, ps)
        rows = torch.arange(0, x.shape[0], 4)
        save("ppi_real_test1_gcn2", dict(row="a1 (config 1, bundled data)", ref="Evaluation/PPI/{ppi_dataloader.py:10-61,"
                                         "gcn_model.py:63-94,train_gcn.py:27,45}", n=int(x.shape[0]), hidden=64,
                                         split="test", graph=1, loader_crosscheck=stats, y_rows="every 4th row"),
             edge_index=ei.to(torch.int32), x=x, labels=y.to(torch.uint8), out_rows=out[rows], out_colsum=out.double().sum(0),
             loss=loss, w0=ps[1], w1=ps[2], w_out=ps[3], b_out=ps[4], grad_x_rows=gs[0][rows], grad_w0=gs[1], grad_w1=gs[2],
             grad_w_out=gs[3], grad_b_out=gs[4])


def gen_fused_pin(ref):
    """Pin for row a10 (FusedKernel/gcn_fused_kernel.cu:5-74, CUDA: cannot be built here): the SAME arithmetic --
    H = relu(A . (X . W)) with unit edge values, duplicates summed, no bias -- is the runnable reference layer
    Evaluation/PPI/gcn_model.py:63-77 (GCNLayer).  Stored: one layer's input, weight, output H and, for the loss
    sum(H * gout), the gradients w.r.t. X and W (what launch_gcn_fused_kernel_backward_optimized must produce)."""
    import tarfile
    import tempfile

    def one(name, edge_index, x, w_scale, seed, extra_meta):
        torch.manual_seed(seed)
        n, Fi = x.shape
        layer = ref.ppi_model.GCNLayer(Fi, 64)
        with torch.no_grad():
            layer.weight.mul_(w_scale)
        xg = x.clone().requires_grad_(True)
        h = layer(edge_index, xg, n)
        gout = torch.randn_like(h)
        gx, gw = torch.autograd.grad(h, [xg, layer.weight], gout)
        save(name, dict(row="a10 (pinned through the runnable GCNLayer)", ref="Evaluation/PPI/gcn_model.py:63-77 == "
                        "FusedKernel/gcn_fused_kernel.cu:39-69 (relu(A.(X.W)), unit values, no bias)", n=int(n), F=int(Fi), H=64,
                        **extra_meta),
             edge_index=edge_index.to(torch.int32), x=x, w=layer.weight, h=h, gout=gout, grad_x=gx, grad_w=gw)

    torch.manual_seed(21)
    n, Fi = 300, 50
    src, dst = rmat_edges(8, 12, seed=5)
    src, dst = src % n, dst % n
    keep = src != dst
    one("fused_gcn_layer_n300", torch.from_numpy(np.vstack((src[keep], dst[keep]))), torch.randn(n, Fi), 0.1, 22,
        dict(duplicate_edges=True))
    with tempfile.TemporaryDirectory() as tmp:
        with tarfile.open(os.path.join(REF, "Evaluation/PPI.tar.xz")) as tf:
            tf.extractall(tmp)
        ei, x, _ = ref.ppi_loader.load_ppi_dataset(os.path.join(tmp, "PPI"), "test")[1]
    one("fused_gcn_layer_ppi_test1", ei, x, 0.05, 23, dict(split="test", graph=1, data="bundled PPI (Evaluation/PPI.tar.xz)"))


def gen_sage_model_grads(ref):
    """a5 with gradients: the same 2-layer GraphSage hop pyramid as sage_model_k3 (same seed, same tensors), plus the
    gradients of every input hop and every parameter for the loss sum(y * gout)."""
    torch.manual_seed(4)
    D, hidden, K, n0 = 12, [16, 8], [3, 3], 10
    model = ref.sageconv.GraphSage(D, hidden, K)
    model.gcn1.reset_parameters()
    model.gcn2.reset_parameters()
    feats = [torch.randn(n0, D), torch.randn(n0 * 3, D), torch.randn(n0 * 9, D)]
    feats = [f.requires_grad_(True) for f in feats]
    y = model(feats)
    gout = torch.randn(y.shape, generator=torch.Generator().manual_seed(41))
    ps = feats + [model.gcn1.weight, model.gcn1.neighborAgg.weight, model.gcn2.weight, model.gcn2.neighborAgg.weight]
    gs = grads_of(y, gout, ps)
    save("sage_model_k3_grads", dict(row="a5", ref="dgll/nn/Convolution/sageconv.py:86-114", D=D, hidden=hidden,
                                     num_neighbors=K, fix=SAGE_FIX, same_tensors_as="sage_model_k3"),
         h0=feats[0], h1=feats[1], h2=feats[2], y=y, gout=gout,
         w1=ps[3], wn1=ps[4], w2=ps[5], wn2=ps[6],
         grad_h0=gs[0], grad_h1=gs[1], grad_h2=gs[2], grad_w1=gs[3], grad_wn1=gs[4], grad_w2=gs[5], grad_wn2=gs[6])


def gen_pooling(ref):
    """Stand-in pin for the global-pooling row (dgll/nn/GlobalPooling/Pooling.py:18-81): the reference calls
    torch_scatter.scatter(x, batch, dim=0, dim_size=size, reduce=...), which is NOT installed in this image and cannot be
    imported.  The vectors below come from torch.Tensor.scatter_reduce (ATen, run in the build container) on seeded
    inputs, with torch_scatter's documented conventions applied by hand: empty segments give 0 for every reduce, 'mean'
    divides by max(count, 1).  The fixture says so in its metadata; it pins the build's kernels to an implementation
    that is independent of them, not to torch_scatter itself."""
    torch.manual_seed(31)
    sizes = [5, 0, 130, 1, 40, 700, 0, 9]                      # two empty graphs, one long segment
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    n, Fd, B = int(batch.numel()), 19, len(sizes)
    x = torch.randn(n, Fd, requires_grad=True)
    idx = batch.view(-1, 1).expand(-1, Fd)
    outs, grads = {}, {}
    for red, name in (("sum", "sum"), ("mean", "mean"), ("amax", "max")):
        y = torch.zeros(B, Fd).scatter_reduce(0, idx, x, red, include_self=False)
        gout = torch.randn(B, Fd, generator=torch.Generator().manual_seed(32))
        (gx,) = torch.autograd.grad(y, x, gout)
        outs[name], grads[name] = y, gx
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(33))     # the same nodes in scrambled order
    save("pooling_scatter_reduce_b8", dict(row="f4 pooling", ref="dgll/nn/GlobalPooling/Pooling.py:18-81", n=n, F=Fd, B=B,
                                           stands_in_for="torch_scatter.scatter (absent from this image): vectors from "
                                           "torch.Tensor.scatter_reduce(include_self=False) on zeros, i.e. empty segments = 0",
                                           gout_seed=32),
         x=x, batch=batch, perm=perm, y_sum=outs["sum"], y_mean=outs["mean"], y_max=outs["max"],
         gout=torch.randn(B, Fd, generator=torch.Generator().manual_seed(32)),
         grad_sum=grads["sum"], grad_mean=grads["mean"], grad_max=grads["max"])


def main():
    ref = load_reference()
    if sys.argv[1:] == ["formats"]:
        return gen_formats(ref)
    if sys.argv[1:] == ["pins"]:                 # round 2: only the new fixtures (the others are unchanged)
        gen_fused_pin(ref)
        gen_sage_model_grads(ref)
        return gen_pooling(ref)
    gen_gcn(ref)
    gen_sage(ref)
    gen_gat(ref)
    gen_ppi(ref)
    gen_sampler(ref)
    gen_adj_prep(ref)
    gen_formats(ref)
    gen_fused_pin(ref)
    gen_sage_model_grads(ref)
    gen_pooling(ref)


if __name__ == "__main__":
    main()
