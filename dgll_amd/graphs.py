"""hipGraph capture of launch-bound training steps.

A full-batch step on a PPI-sized graph (BASELINE config 1: ~2 k nodes) is ~60 kernel launches of a few microseconds each;
issued one by one from Python it takes ~0.8 ms of which the GPU is busy a fraction.  The step has static shapes, so it is
captured ONCE into a HIP graph -- the library's launches go to torch's current stream, which is the capturing stream, so
they are recorded like torch's own -- and replayed with a single host call (`torch.cuda.CUDAGraph` = hipGraph on ROCm).

Everything that is not a stream operation must have happened before capture: the CSR build, its launch plan (hipMalloc),
the transposed CSR for the backward pass -- the warm-up iterations run on a side stream first take care of that.  The
optimizer must be capture-safe (`torch.optim.Adam(..., capturable=True)`).  The reference has nothing comparable (its loop
is eager, Evaluation/PPI/train_gcn.py:30-53)."""
import torch


class GraphedTrainStep:
    """Captured `zero_grad -> forward -> loss -> backward -> optimizer.step` for FIXED input tensors.

    step_fn() must run the forward pass and return the scalar loss, reading only tensors that stay alive and in place
    (e.g. `lambda: crit(model(edge_index, x), y)`).  Pass a LIST of such functions to capture a whole epoch of steps
    (one per training graph, executed in list order) into ONE HIP graph: a single host call then replays the epoch;
    `losses` holds one static scalar per step."""

    def __init__(self, step_fn, optimizer, warmup=3, sync_after_replay=False):
        self.sync_after_replay = sync_after_replay
        step_fns = list(step_fn) if isinstance(step_fn, (list, tuple)) else [step_fn]

        def step_fn_all(record=None):
            for fn in step_fns:
                optimizer.zero_grad(set_to_none=False)
                loss = fn()
                loss.backward()
                optimizer.step()
                if record is not None:
                    record.append(loss.detach())

        if not torch.cuda.is_available():
            raise RuntimeError("GraphedTrainStep captures a HIP graph: a GPU is required")
        for group in optimizer.param_groups:
            if "capturable" in group and not group["capturable"]:
                raise ValueError("the optimizer must be constructed with capturable=True to be captured in a HIP graph")
        self.optimizer = optimizer
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        # Warm-up AND capture run on the same side stream: autograd pins each parameter's gradient accumulation to the
        # stream of the parameter's first use; capturing on another stream would fork to it and join back for every
        # parameter (torch warns about exactly that).
        with torch.cuda.stream(side):                      # warm-up: builds CSRs, plans, transposes, autotunes the GEMMs
            for group in optimizer.param_groups:   # gradients in ordinary tensors, zeroed inside the graph (below)
                for p in group["params"]:
                    if p.requires_grad and p.grad is None:
                        p.grad = torch.zeros_like(p)
            for _ in range(warmup):
                step_fn_all()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self.losses = []
        with torch.cuda.graph(self.graph, stream=side):
            step_fn_all(self.losses)             # static outputs; no autograd graph is kept alive between replays
            self.total = torch.stack(self.losses).sum()      # the epoch's loss, computed inside the graph
        self.loss = self.losses[-1]

    def __call__(self):
        """Replay; returns the (static) loss tensor of the last step.  `losses` (one per step) and `total` (their sum,
        computed inside the graph) are static too: read them before the next replay.

        KNOWN HAZARD on this stack (ROCm 7.2, torch 2.10+rocm7.0), independent of this library (reproduced with a
        pure-torch MLP, tools/graph_replay_check.py): a torch reduction of a large tensor to ONE value inside the captured
        step -- e.g. nn.CrossEntropyLoss's mean over [n, classes] -- uses a multi-block kernel whose semaphore buffer is
        cleared by a captured memset node; when other device work runs between replays those reductions come back as 0
        on later replays (the parameters still train bit-identically to the eager loop; only the reported scalar is
        wrong).  Two-stage reductions are unaffected: use `dgll_amd.ops.cross_entropy` (row losses from one kernel, summed
        as [n] -> [n/256] -> scalar) or reduce per row first."""
        self.graph.replay()
        if self.sync_after_replay:
            torch.cuda.current_stream().synchronize()
        return self.loss


class PaddedBlock:
    """Factory of the CSR block of one hop of a sampled batch on STATIC shapes: `rows` destination rows over `cols` source rows
    (cols = rows x fan-out, the hop's upper bound), the row pointers in a buffer that is refilled per batch (pad(): the rows a
    batch does not use are empty, the source rows it does not use belong to no edge).  Everything derived from the row pointers
    is recomputed on every call -- inside a captured step that means on every replay.  The kernels' launch plan of a padded block is
    the host-only one whatever the fan-out (graph.CSRGraph.plan: no long-row chunk schedule, which would describe the capture-time
    row pointers and go stale on the first replay; rows of any length are gathered inline)."""

    @staticmethod
    def make(rows, fanout, device, cols=None):
        """cols: bound on the block's edges (= rows of the next hop); default rows x fanout, the hop's upper bound."""
        from .graph import CSRGraph

        class _Block(CSRGraph):
            identity_cols = True
            padded = True

            def degrees(self):                       # never cached: the row pointers change under the captured step
                return self.rowptr[1:] - self.rowptr[:-1]

            def row_index(self):
                """Destination row of every source row; n_rows for the source rows past the batch's edges (they have none: the block's
                backward reads an appended all-zero gradient row for them, ops._Spmm.backward)."""
                return torch.searchsorted(self.rowptr[1:], self._edge_ids, right=True)

        cols = rows * fanout if cols is None else int(cols)
        rowptr = torch.clamp(torch.arange(0, rows * fanout + 1, fanout, dtype=torch.int64, device=device), max=cols)
        g = _Block(rowptr, torch.arange(cols, dtype=torch.int32, device=device), None, rows, cols, check=False)
        g.max_degree = fanout
        g._edge_ids = torch.arange(cols, dtype=torch.int64, device=device)
        return g

    @staticmethod
    def pad(block, rowptr, n_edges):
        """Load a batch's row pointers ([n + 1] entries, the last one = n_edges) into the block's static buffer."""
        n = int(rowptr.numel()) - 1
        if n > block.n_rows or n_edges > block.n_cols:
            raise ValueError("a batch larger than the static block (%d rows / %d edges against %d / %d)" % (n, n_edges, block.n_rows,
                                                                                                         block.n_cols))
        block.rowptr[:n + 1].copy_(rowptr, non_blocking=True)
        if n < block.n_rows:
            block.rowptr[n + 1:].fill_(n_edges)


class GraphedSampledStep:
    """forward_sampled -> cross-entropy -> backward of a sampled GraphSAGE batch as ONE HIP graph, the optimizer step behind it.

    A sampled step on the Reddit shape is ~80 launches of a few microseconds each: issued one by one through Python the consumer
    thread needs 2.1 ms per batch for 1.4 ms of kernels, and it shares the interpreter with the loading thread.  The blocks of a
    batch differ in size from batch to batch but are bounded by batch x prod(fan-outs): the step is captured ONCE on those bounds
    (PaddedBlock) and replayed on every batch after a handful of copies into its static inputs.  Rows a batch does not fill are
    empty rows / rows no edge points at: they receive zero gradient (their labels are ignored, no edge carries gradient to
    them: the block's backward reads an all-zero row for them) and hold finite stale values, so they add exact zeros to the weight gradients.  The optimizer's step stays outside the
    graph (its bias corrections are host scalars of the launch): one more launch.

    model: GraphSage on the GPU (standard layers: hops of a layer batched, forward_sampled's fast path); optimizer: FlatAdam (the
    weight-gradient kernels write its gradient slots in place -- the replay refills them) or any torch optimizer.  Every captured
    set keeps the gradient tensors ITS graph writes (a FlatAdam slot is the same tensor in every set; anything else -- a torch
    optimizer's accumulated gradients, a parameter with two producers -- lives in that graph's pool) and p.grad is bound to them
    before the optimizer steps: without that, the replay of set k < n_sets - 1 would fill its own tensors while the optimizer read
    the ones the LAST capture left in p.grad.
    fanouts: the sampler's, in the model's order; the outermost hop arrives reduced (Batch.last_hop_reduced)."""

    def __init__(self, model, optimizer, batch_size, fanouts, in_feats, n_classes, dtype=torch.bfloat16, device="cuda", warmup=2,
                 rows=None, n_sets=1):
        """n_sets > 1: that many independent sets of static inputs, each with its own captured graph.  Set 0 is the one `load()` copies
        into; sets 1 .. n_sets-1 are written IN PLACE by the pipeline's loading stage (MiniBatchPipeline.use_static_sets: the cache
        gather, the outermost hop's reduction, the row pointers and the labels of a batch land directly in a set; the consumer
        replays that set's graph without copying anything -- at the Reddit shape the copies were 260 MB per batch) and handed
        round: a set is rewritten only after the replay that read it has passed (`free` event)."""
        from . import ops

        if not torch.cuda.is_available():
            raise RuntimeError("GraphedSampledStep captures a HIP graph: a GPU is required")
        self.model, self.optimizer = model, optimizer
        dev = self.device = torch.device(device)
        order = [int(f) for f in reversed(fanouts)]                  # hop h -> h + 1 was sampled with order[h] (base_sampler.py:30-58)
        L = self.L = len(order)
        self.rows = [int(batch_size)]
        for f in order[:-1]:
            self.rows.append(self.rows[-1] * f)                      # rows of hops 0 .. L-1 (upper bounds)
        if rows is not None:
            # tighter bounds chosen by the caller (what its batches actually reach, with a margin): the padding is GPU work -- on a
            # graph with many low-degree nodes the hops fill 70 % of batch x prod(fan-outs) and the upper bounds cost 40 % more
            # kernel time than the batches need.  A batch that exceeds them is refused by load(): run it launch by launch.
            if len(rows) != L or any(int(r) < 1 for r in rows):
                raise ValueError("rows: one bound per hop 0 .. L-1")
            self.rows = [min(int(r), cap) for r, cap in zip(rows, self.rows)]
        self._ops = ops
        self.sets = []
        for _ in range(max(1, int(n_sets))):
            self.sets.append(self._make_set(order, in_feats, n_classes, dtype, warmup))
        first = self.sets[0]                                          # the copy path's set under the names earlier rounds used
        self.feat_all, self.features, self.agg_all, self.reduced = first.feat_all, first.features, first.agg_all, first.reduced
        self.blocks, self.labels, self.graph, self.loss = first.blocks, first.labels, first.graph, first.loss

    def _make_set(self, order, in_feats, n_classes, dtype, warmup):
        import types

        ops, dev, L = self._ops, self.device, self.L
        st = types.SimpleNamespace()
        total = sum(self.rows)
        store = ops.alloc_features(total, in_feats, dtype, dev)
        store.zero_()
        st.feat_all = store
        offs = [0]
        for r in self.rows:
            offs.append(offs[-1] + r)
        st.features = [store[offs[h]:offs[h + 1]] for h in range(L)]               # consecutive row slices: stacked without a copy
        # the first layer's neighbour reductions of all hops in one buffer: the outermost hop's (it arrives ready-made, per hop L-1
        # row) is its tail, the model writes the others in front of it (GraphSage._forward_sampled_batched) -- no concatenation
        st.agg_all = ops.alloc_features(total, in_feats, dtype, dev)
        st.agg_all.zero_()
        st.reduced = st.agg_all[offs[L - 1]:offs[L]]
        st.reduced._dgll_stack = st.agg_all
        st.blocks = [PaddedBlock.make(self.rows[h], order[h], dev, cols=self.rows[h + 1]) for h in range(L - 1)] + [None]
        st.labels = torch.full((self.rows[0],), -100, dtype=torch.int64, device=dev)
        st.free = None                 # event behind the last replay that read this set (the loading stage waits for it)
        # synthetic full-size contents for the warm-up and the capture (finite features, every label valid)
        store.normal_()
        st.reduced.normal_()
        st.labels.random_(0, int(n_classes))
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                                 # same stream for warm-up and capture (see GraphedTrainStep)
            for _ in range(max(1, warmup)):
                self._zero_grad()
                self._forward_backward(st)
            self._zero_grad()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        st.graph = torch.cuda.CUDAGraph()
        # one memory pool for all sets: the replays never overlap (one stream), so only the static inputs and what a set hands out
        # (its loss, its gradient tensors -- kept alive below) need to be per set; a pool per set reserved the whole activation
        # footprint n_sets times and took it from the feature cache
        pool = self.sets[0].graph.pool() if getattr(self, "sets", None) else None
        with torch.cuda.graph(st.graph, pool=pool, stream=side, capture_error_mode="thread_local"):
            st.loss = self._forward_backward(st)
        torch.cuda.synchronize(dev)
        st.params = [p for p in self.model.parameters() if p.requires_grad]
        st.grads = [p.grad for p in st.params]          # the tensors THIS graph's replay fills
        return st

    def _bind_grads(self, st):
        for p, g in zip(st.params, st.grads):
            if p.grad is not g:
                p.grad = g

    def _zero_grad(self):
        self.optimizer.zero_grad(set_to_none=True)

    def _forward_backward(self, st=None):
        st = self.sets[0] if st is None else st
        out = self.model.forward_sampled(st.features, st.blocks, last_hop_reduced=st.reduced)
        loss = self._ops.cross_entropy(out, st.labels)
        loss.backward()
        return loss.detach()

    def load(self, batch):
        """Copy one pipeline batch (features of hops 0 .. L-1, the reduced outermost hop, CSR blocks, labels) into the static inputs
        of set 0, on the current stream."""
        L = self.L
        n = [int(batch.features[h].shape[0]) for h in range(L)]
        for h in range(L):
            if n[h] > self.rows[h]:
                raise ValueError("hop %d of the batch has %d rows, the captured step holds %d" % (h, n[h], self.rows[h]))
            self.features[h][:n[h]].copy_(batch.features[h], non_blocking=True)
        self.reduced[:n[L - 1]].copy_(batch.last_hop_reduced, non_blocking=True)
        for h in range(L - 1):
            PaddedBlock.pad(self.blocks[h], batch.blocks[h].rowptr, n[h + 1])
        if n[0] < self.rows[0]:
            self.labels.fill_(-100)
        self.labels[:n[0]].copy_(batch.labels, non_blocking=True)

    def eager(self, batch):
        """The same step launch by launch (diagnostics: per-launch events) on the batch's own tensors -- or, for a batch that was
        loaded in place, on its static set (padded shapes)."""
        k = getattr(batch, "static_set", None)
        if k is None:
            out = self.model.forward_sampled(batch.features, batch.blocks, last_hop_reduced=batch.last_hop_reduced)
            loss = self._ops.cross_entropy(out, batch.labels)
        else:
            st = self.sets[k]
            out = self.model.forward_sampled(st.features, st.blocks, last_hop_reduced=st.reduced)
            loss = self._ops.cross_entropy(out, st.labels)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self._mark_free(k)
        self.optimizer.step()
        return loss.detach()

    def _mark_free(self, k):
        if k is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.sets[k].free = ev

    def __call__(self, batch):
        """One training step on `batch`; returns the (static) loss tensor: read it before the next call.  A batch the loading stage
        wrote in place (batch.static_set = k) replays set k's graph as it is; any other batch is copied into set 0 first."""
        from .ranges import rng

        k = getattr(batch, "static_set", None)
        with rng("consume"):
            if k is None:
                self.load(batch)
                st = self.sets[0]
            else:
                st = self.sets[k]
            st.graph.replay()
            self._mark_free(k)
            self._bind_grads(st)
            self.optimizer.step()
        return st.loss
