"""GraphSAGE layer on the gfx950 aggregation engine.

Interface mirror of /root/reference/dgll/nn/Convolution/sageconv.py:
  NeighborAggregator(input_dim, output_dim, use_bias=False, aggr_method="mean")      sageconv.py:10-45
  sageConv(input_dim, hidden_dim, activation=relu, aggr_neighbor_method="mean",
           aggr_hid_method="sum")                                                    sageconv.py:48-83
  GraphSage(input_dim, hidden_dim=[64, 64], num_neighbors_list=[10, 10])             sageconv.py:86-114
with the same parameter names (`weight`, `neighborAgg.weight`, `neighborAgg.bias`, `gcn1`, `gcn2`, ...).

Reference defects deliberately NOT reproduced: the K-axis reduction result is
assigned (sageconv.py:33-38 discards it), `max` takes the values of torch.max, sageConv.weight is
initialised (reset_parameters() is never called at sageconv.py:63-68), GraphSage accepts any number of
layers and indexes fan-outs by hop (sageconv.py:111 indexes by layer, which only works for equal fan-outs).

On the GPU the K-axis reduce over the dense [N, K, D] neighbour tensor is the CSR SpMM kernel on a
constant-degree structure (rowptr = arange(0, N*K+1, K), col = arange(N*K)); `forward_block` takes a sampled
CSR block and gathers straight from the source feature matrix without materialising [N, K, D].
"""
from ... import backend as F
from ... import dense, ops
from ...graph import CSRGraph

_fanout_graphs = {}


def _fanout_graph(n, k, device):
    key = (n, k, str(device))
    g = _fanout_graphs.get(key)
    if g is None:
        if len(_fanout_graphs) > 64:
            _fanout_graphs.clear()
        g = _fanout_graphs[key] = CSRGraph.fixed_fanout(n, k, device)
    return g


def _check_method(name, allowed, what):
    if name not in allowed:
        raise ValueError("Unsupported %s, expected %s, but got %s" % (what, ", ".join(allowed), name))


class NeighborAggregator(F.nn.Module):
    """reduce_K(neighbours) . weight (+ bias)."""

    def __init__(self, input_dim, output_dim, use_bias=False, aggr_method="mean"):
        super().__init__()
        self.input_dim, self.output_dim = input_dim, output_dim
        self.use_bias, self.aggr_method = use_bias, aggr_method
        self.weight = F.Parameter(F.empty(input_dim, output_dim))
        if use_bias:
            self.bias = F.Parameter(F.empty(output_dim))
        self.reset_parameters()

    def reset_parameters(self):
        F.init.kaiming_uniform_(self.weight)        # sageconv.py:28
        if self.use_bias:
            F.init.zeros_(self.bias)                # sageconv.py:30

    def reduce(self, neighbor_feature):
        """[N, K, D] -> [N, D] over the K axis (sageconv.py:33-38, with the result kept)."""
        _check_method(self.aggr_method, ("mean", "sum", "max"), "aggr_method")
        if not neighbor_feature.is_cuda:             # CPU tensors: torch's own reduction, as in the reference
            if self.aggr_method == "max":
                return neighbor_feature.max(dim=1)[0]
            return getattr(neighbor_feature, self.aggr_method)(dim=1)
        n, k, d = neighbor_feature.shape
        flat = neighbor_feature.reshape(n * k, d)
        graph = _fanout_graph(n, k, neighbor_feature.device)
        if self.aggr_method == "max":
            return ops.segment_max(graph, flat)
        return ops.spmm(graph, flat, reduce=self.aggr_method)

    def reduce_block(self, block, x_src, out=None):
        """Same reduction driven by a CSR block (rows = destination nodes, columns index x_src).  out: rows of a caller's buffer to
        write (mean / sum on the GPU)."""
        _check_method(self.aggr_method, ("mean", "sum", "max"), "aggr_method")
        if self.aggr_method == "max":
            return ops.segment_max(block, x_src)
        if out is not None and x_src.is_cuda:
            return ops.spmm(block, x_src, reduce=self.aggr_method, out=out)
        return ops.spmm(block, x_src, reduce=self.aggr_method)

    def transform(self, reduced):
        hidden = F.matmul(reduced, self.weight.to(reduced.dtype))     # sageconv.py:41
        if self.use_bias:
            hidden = hidden + self.bias.to(hidden.dtype)
        return hidden

    def forward(self, neighbor_feature):
        return self.transform(self.reduce(neighbor_feature))


class sageConv(F.nn.Module):
    """act( src . weight  (+ | ++)  NeighborAggregator(neighbours) ) -- sageconv.py:70-83."""

    def __init__(self, input_dim, hidden_dim, activation=F.relu, aggr_neighbor_method="mean", aggr_hid_method="sum"):
        super().__init__()
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        self.activation = activation
        self.aggr_neighbor_method, self.aggr_hid_method = aggr_neighbor_method, aggr_hid_method
        self.weight = F.Parameter(F.empty(input_dim, hidden_dim))
        self.neighborAgg = NeighborAggregator(input_dim, hidden_dim, aggr_method=aggr_neighbor_method)
        self.reset_parameters()

    reorder = True   # allow transform-before-aggregate in forward_block when the layer narrows

    def reset_parameters(self):
        F.init.kaiming_uniform_(self.weight)        # sageconv.py:68

    def _combine(self, self_hidden, neighbor_hidden):
        _check_method(self.aggr_hid_method, ("sum", "concat"), "aggr_hid_method")
        if self.aggr_hid_method == "sum":
            hidden = self_hidden + neighbor_hidden
        else:
            hidden = F.cat([self_hidden, neighbor_hidden], dim=1)   # self first (sageconv.py:77)
        return self.activation(hidden) if self.activation else hidden

    def forward(self, src_node_features, neighbor_node_features):
        neighbor_hidden = self.neighborAgg(neighbor_node_features)
        self_hidden = F.matmul(src_node_features, self.weight.to(src_node_features.dtype))
        return self._combine(self_hidden, neighbor_hidden)

    def forward_block(self, block, x_src, x_dst=None):
        """Sampled-block / full-graph form: `block` is a CSRGraph whose rows are the destination nodes and whose
        columns index `x_src`; `x_dst` are the destination nodes' own features (default: the first n_rows rows
        of x_src, the usual block convention)."""
        if x_dst is None:
            x_dst = x_src[:block.n_rows]
        # A sampled block whose source rows each belong to exactly ONE edge (identity_cols: the reference's neighbour lists keep
        # duplicates, dgllsampler.py:17) is aggregated FIRST whatever the widths: the reduction shrinks the row count by the
        # fan-out before any product, and the weight gradient then runs over the destination rows, not over fan-out times as many
        # sources.  (The narrow-product-first order pays when source rows are shared by many edges: the full graph.)
        if self.transform_first(x_src) and not getattr(block, "identity_cols", False):
            # mean/sum are linear: reduce(X).W_n == reduce(X.W_n).  When the layer narrows (hidden < input) aggregate
            # the NARROW product -- fewer bytes per gathered edge in the forward and in the backward gather.
            z = dense.linear(x_src, self.neighborAgg.weight)
            return self.finish_transform_first(x_dst, self.neighborAgg.reduce_block(block, z))
        return self.transform_block(x_dst, self.neighborAgg.reduce_block(block, x_src))

    def transform_first(self, x):
        """Whether forward_block applies W_n before the neighbour reduction (set sageConv.reorder = False to keep the
        reference's aggregate-then-transform order, sageconv.py:33-41, everywhere)."""
        return (self.reorder and x.is_cuda and self.hidden_dim < self.input_dim
                and self.aggr_neighbor_method in ("mean", "sum") and self.aggr_hid_method == "sum"
                and not self.neighborAgg.use_bias and self.activation in (None, F.relu))

    def finish_transform_first(self, x_dst, reduced_z):
        return dense.add_linear_act(reduced_z, x_dst, self.weight, self.activation is not None)

    def transform_block(self, x_dst, reduced):
        """act(x_dst . weight (+|++) reduced . neighborAgg.weight) given the already aggregated neighbours."""
        if (x_dst.is_cuda and self.aggr_hid_method == "sum" and not self.neighborAgg.use_bias
                and self.activation in (None, F.relu)):
            return dense.sage_transform(x_dst, reduced, self.weight, self.neighborAgg.weight, self.activation is not None)
        return self._combine(F.matmul(x_dst, self.weight.to(x_dst.dtype)), self.neighborAgg.transform(reduced))


class GraphSage(F.nn.Module):
    """Hop-pyramid GraphSAGE (sageconv.py:103-114): layer l is applied to hops 0 .. L-l-1, hop h taking its
    neighbours from hop h+1 viewed as [n_h, K_h, -1]."""

    def __init__(self, input_dim, hidden_dim=[64, 64], num_neighbors_list=[10, 10]):
        super().__init__()
        self.input_dim, self.hidden_dim = input_dim, list(hidden_dim)
        # None: full-graph use only (forward_graph), one layer per hidden_dim entry
        self.num_neighbors_list = [None] * len(self.hidden_dim) if num_neighbors_list is None else list(num_neighbors_list)
        self.num_layers = len(self.num_neighbors_list)
        if len(self.hidden_dim) != self.num_layers:
            raise ValueError("hidden_dim and num_neighbors_list must have one entry per layer")
        dims = [input_dim] + self.hidden_dim
        self.gcn = []
        for l in range(self.num_layers):
            layer = sageConv(dims[l], dims[l + 1])
            setattr(self, "gcn%d" % (l + 1), layer)   # registered as gcn1, gcn2, ... (sageconv.py:96-97)
            self.gcn.append(layer)

    def forward(self, node_feature_list):
        hidden = node_feature_list
        for l in range(self.num_layers):
            layer = self.gcn[l]
            hidden = [layer(hidden[hop], hidden[hop + 1].view(len(hidden[hop]), self.num_neighbors_list[hop], -1))
                      for hop in range(self.num_layers - l)]
        return hidden[0]

    def forward_sampled(self, node_feature_list, blocks, last_hop_reduced=None):
        """Hop pyramid driven by sampled CSR blocks (variable fan-out, e.g. DGLLNeighborSampler): blocks[hop] has one row
        per node of hop `hop` and gathers from the rows of hop `hop + 1` (sugbraph.to_block()).
        last_hop_reduced: the first layer's neighbour reduction over the OUTERMOST block, already formed (by
        GraphCacheServer.aggregate_data through MiniBatchPipeline(reduce_last_hop=...)): node_feature_list[-1] is then not
        needed (None) -- those rows enter the model through this reduction only (sageconv.py:33-36)."""
        hidden = list(node_feature_list)
        L = self.num_layers
        if self._hops_batchable(hidden, blocks, last_hop_reduced):
            return self._forward_sampled_batched(hidden, blocks, last_hop_reduced)
        for l in range(L):
            layer = self.gcn[l]
            nxt = []
            for hop in range(L - l):
                if l == 0 and hop == L - 1 and last_hop_reduced is not None:
                    nxt.append(layer.transform_block(hidden[hop], last_hop_reduced))
                else:
                    nxt.append(layer.forward_block(blocks[hop], hidden[hop + 1], hidden[hop]))
            hidden = nxt
        return hidden[0]

    batch_hops = True      # forward_sampled: one transform / one pair of weight gradients per LAYER instead of per (layer, hop)

    def _hops_batchable(self, hidden, blocks, last_hop_reduced):
        """Every hop of a layer goes through the same weights: on the GPU, with the standard layer configuration and blocks that
        are aggregated before the transform, the hops' rows are stacked and a layer is ONE transform forward and ONE pair of
        weight-gradient launches backward (a 3-layer model on 3 hops: 6 transforms and 12-16 weight-gradient launches become 3
        and 3-4; each of them was at the ~20 us floor of a launch on a few thousand rows)."""
        first = hidden[0]
        if not (self.batch_hops and first.is_cuda and self.num_layers >= 2):
            return False
        for l, layer in enumerate(self.gcn):
            if (layer.aggr_hid_method != "sum" or layer.neighborAgg.use_bias or layer.aggr_neighbor_method not in ("mean", "sum")
                    or layer.activation not in (None, F.relu)):
                return False
        # every block is reduced BEFORE the transform (identity columns), or the layer does not narrow
        L = self.num_layers      # (the outermost block may be None when its reduction arrives ready-made: it is never touched)
        return all((b is None and i == L - 1 and last_hop_reduced is not None) or getattr(b, "identity_cols", False)
                   for i, b in enumerate(blocks[:L]))

    @staticmethod
    def _stack_rows(parts):
        """One [sum rows, F] tensor holding the parts' rows in order: the parts themselves when they already are consecutive row
        slices of one buffer (the pipeline fetches the hops with one gather), else a copy."""
        if len(parts) == 1:
            return parts[0]
        a = parts[0]
        esz, ld = a.element_size(), a.stride(0)
        adjacent = a.dim() == 2 and a.stride(1) == 1 and not a.requires_grad
        ptr, base = a.data_ptr(), a.untyped_storage().data_ptr()
        for t in parts:     # same storage, same pitch, one right behind the other
            adjacent = adjacent and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) == ld and t.shape[1] == a.shape[1] \
                and t.data_ptr() == ptr and t.untyped_storage().data_ptr() == base and not t.requires_grad and t.dtype == a.dtype
            ptr += t.shape[0] * ld * esz
        if adjacent:
            return a.as_strided((sum(t.shape[0] for t in parts), a.shape[1]), (ld, 1))
        return F.cat(list(parts), dim=0)

    def _forward_sampled_batched(self, hidden, blocks, last_hop_reduced):
        L = self.num_layers
        sizes = [h.shape[0] for h in hidden[:L]]                      # rows of hops 0 .. L-1 (hop L only feeds a reduction)
        parent = None                                                 # rows of hops 0 .. n_h of the current layer's input, stacked
        for l in range(L):
            layer = self.gcn[l]
            n_h = L - l                                               # destination hops 0 .. n_h - 1
            offs = [0]
            for hop in range(n_h + (1 if l > 0 else 0)):
                offs.append(offs[-1] + sizes[hop])
            if l == 0:
                src = lambda hop: hidden[hop]                                                    # noqa: E731
            else:
                # the row ranges of the layer's input this layer reads: [0, offs[n_h]) as the transform's self operand and one range per
                # block's sources -- taken through ONE autograd node (ops.row_slices) whose backward assembles their gradients in a
                # single buffer; as plain slices every range came back as a full-size zero-filled tensor, summed pairwise
                bounds = [(0, offs[n_h])] + [(offs[hop + 1], offs[hop + 2]) for hop in range(n_h)]
                views = ops.row_slices(parent, bounds)
                x_dst_view, hop_rows = views[0], views[1:]                                       # hop_rows[k]: the rows of hop k + 1
                src = lambda hop, hop_rows=hop_rows: hop_rows[hop - 1]                           # noqa: E731
            # the first layer's reductions in ONE buffer when the ready-made outermost one already sits at its tail (the tensor
            # carries `_dgll_stack`, graphs.GraphedSampledStep): the hops' reductions are written in front of it and the stacked
            # operand of the transform needs no concatenation (a 130 MB copy per batch at the Reddit shape)
            stack = None
            if l == 0 and last_hop_reduced is not None and layer.aggr_neighbor_method in ("mean", "sum"):
                stack = getattr(last_hop_reduced, "_dgll_stack", None)
                if stack is not None and not (stack.shape[0] == offs[n_h] and stack.dtype == last_hop_reduced.dtype
                                              and stack[offs[L - 1]:].data_ptr() == last_hop_reduced.data_ptr()):
                    stack = None
            aggs = []
            for hop in range(n_h):
                if l == 0 and hop == L - 1 and last_hop_reduced is not None:
                    aggs.append(last_hop_reduced)
                else:
                    aggs.append(layer.neighborAgg.reduce_block(blocks[hop], src(hop + 1),
                                                               out=stack[offs[hop]:offs[hop + 1]] if stack is not None else None))
            x_dst = self._stack_rows([hidden[hop] for hop in range(n_h)]) if l == 0 else x_dst_view
            parent = layer.transform_block(x_dst, self._stack_rows(aggs))     # hops 0 .. n_h - 1 of the next layer's input
        return parent
    def forward_graph(self, graph, x):
        """Full-graph form (BASELINE configs 3 and 5): every layer aggregates over the whole adjacency `graph`
        (a CSRGraph), h <- layer(h_self = h, neighbours of each node gathered from h)."""
        from ... import fused_layers

        fusable = [fused_layers.can_fuse(layer, x.is_cuda) for layer in self.gcn]
        # layer i hands the gradient of its input back already masked by layer i-1's ReLU (written by the epilogue of the
        # kernel that produces it); layer i-1 then skips its own masking pass.  Only between two fused layers.
        gates = [i > 0 and fusable[i] and fusable[i - 1] and self.gcn[i - 1].activation is not None
                 for i in range(len(self.gcn))]
        h = x
        for i, layer in enumerate(self.gcn):
            out = fused_layers.sage_graph_layer(layer, graph, h, gate_input=gates[i],
                                                grad_is_gated=i + 1 < len(gates) and gates[i + 1])
            h = layer.forward_block(graph, h, h) if out is None else out
        return h
