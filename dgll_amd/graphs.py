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
