"""FlatAdam -- the optimizer step of the training loops as ONE launch over flat parameter / gradient buffers.

The reference's loops run torch.optim.Adam(model.parameters()) after a per-parameter all-reduce (GPU Accelerator/MQGCN.py:55-79,
141-144; Evaluation/PPI/train_gcn.py:26).  On the GPU that is seven multi-tensor launches per step, plus -- with RaCoM's flattened
bucket (dist.RaCoM) -- one copy per parameter into the bucket and one back, plus one pack launch per weight matrix and product
(dense._pad_wt).  At 8 ranks of the products-sized graph a rank's whole step is ~4 ms and these ~40 launches were a tenth of it.

Here:
  * every parameter is a view of ONE fp32 buffer (`flat`), every gradient a view of a second one (`grad`);
  * the layers' weight-gradient kernels write their result straight into the parameter's gradient slot (`grad_slot`; autograd then
    adopts that view as `.grad` without a copy), so the bucket RaCoM all-reduces IS the gradients: no copy in, no copy out;
  * `step()` is one kernel (dgll_hip_adam_flat: torch.optim.Adam's arithmetic) that also emits, for the weight matrices of the
    MFMA transforms, the two packed bf16 forms those kernels take (`pack_weights=True`): dense._pad_wt finds them here instead
    of launching a pack per product.

Contract of the packed forms: they follow every update made through this optimizer and through ordinary in-place tensor ops (the
parameter's version counter is checked on every lookup; a mismatch re-packs).  Writes through `param.data` are invisible to the
counter -- call `repack()` after such a write, or construct with pack_weights=False.
"""
import ctypes as C

import torch

from . import _lib


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam over flat buffers.  A torch Optimizer (param_groups with lr / betas / eps / weight_decay, read on EVERY step:
    learning-rate schedulers work; `params` may be a list of parameters or of parameter-group dicts), with torch.optim.Adam's
    behaviour for a parameter WITHOUT a gradient at step(): it is skipped entirely -- value, both moments and its own step count
    untouched, no weight decay (round 4 zero-filled its slot: stale momentum kept moving it).  The common case -- every parameter
    has a gradient -- is one launch; parameters that were skipped at some point (their step counts, hence bias corrections, differ)
    or groups with different hyper-parameters are updated by one launch per run of consecutive parameters.
    Not capturable: step() computes the bias corrections on the host per call and refuses to run under stream capture."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, pack_weights=True):
        params = list(params)
        super().__init__(params, dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps), weight_decay=float(weight_decay)))
        self.params, self._group_of = [], []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.requires_grad:
                    self.params.append(p)
                    self._group_of.append(gi)
        if not self.params:
            raise ValueError("FlatAdam got no parameter that requires grad")
        dev = self.params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise ValueError("FlatAdam takes fp32 parameters on one device")
        self.device = dev
        self.steps = 0                 # step() calls so far; param_steps: updates every parameter has received (torch's state['step'])
        self.param_steps = [0] * len(self.params)
        self.grad_scale = 1.0          # ONE-SHOT factor on the gradients of the next step(): RaCoM sets 1 / world_size (the reference's
                                       # average, MQGCN.py:64) instead of a div_ launch; clip_grad_norm_ folds its coefficient in.  step()
                                       # consumes it and resets it to 1.0
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += -(-p.numel() // 4) * 4                      # every slot starts on a 16-byte boundary
        self.total = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                self.flat[o:o + p.numel()].copy_(p.detach().reshape(-1))
                p.data = self.flat[o:o + p.numel()].view(p.shape)            # the parameter now lives in the flat buffer
                p._dgll_flat = (self, i)
        self._slot_of = {id(p): i for i, p in enumerate(self.params)}
        self._handed_out = [False] * len(self.params)       # a slot is given to ONE producer per accumulation window
        self.pack_weights = bool(pack_weights) and dev.type == "cuda"
        self._packed = {}              # param index -> (packed W, packed W^T)
        self._seen_version = {}
        self._table = None
        if self.pack_weights:
            self.repack()

    # hyper-parameters of the first group under their old attribute names (read-only convenience; the groups are what step() reads)
    lr = property(lambda self: self.param_groups[0]["lr"])
    betas = property(lambda self: self.param_groups[0]["betas"])
    eps = property(lambda self: self.param_groups[0]["eps"])
    weight_decay = property(lambda self: self.param_groups[0]["weight_decay"])

    # ---- gradient slots --------------------------------------------------------------------------------------------
    def grad_slot(self, p):
        """A FRESH [shape of p] view of p's slice of the gradient buffer: hand it to a kernel as its output and return it from an
        autograd backward -- AccumulateGrad adopts a tensor nobody else references as .grad without copying it."""
        i = self._slot_of[id(p)]
        o = self.offsets[i]
        return self.grad[o:o + p.numel()].view(p.shape)

    def claim_slot(self, p):
        """grad_slot(p) for a producer inside a backward pass -- or None when the slot was already claimed since the last
        zero_grad() / step() (a parameter used by two nodes, or gradient accumulation over several backward passes: the second
        producer must return a tensor of its own, which autograd then ADDS to the first)."""
        i = self._slot_of[id(p)]
        if self._handed_out[i] or p.grad is not None:
            return None
        self._handed_out[i] = True
        return self.grad_slot(p)

    def _slot_ptr(self, i):
        return self.grad.data_ptr() + 4 * self.offsets[i]

    def gather_grads(self, zero_missing=True):
        """Make the gradient buffer hold every parameter's gradient: slots written in place are left alone, a gradient that autograd
        produced elsewhere is copied in.  zero_missing (what a gradient all-reduce over the whole buffer needs, dist.RaCoM: another
        rank may hold a gradient for it): a parameter without gradient gets a zero slot as its .grad; False (step()): it is left
        without one and the update skips it."""
        for i, p in enumerate(self.params):
            g = p.grad
            if g is not None and g.data_ptr() == self._slot_ptr(i) and g.is_contiguous():
                continue
            if g is None and not zero_missing:
                continue
            slot = self.grad_slot(p)
            if g is None:
                slot.zero_()
            else:
                slot.copy_(g)
            p.grad = slot

    def grad_norm(self):
        """2-norm of the gradients as the next step() will apply them (grad_scale included: after an in-place RaCoM reduction .grad
        holds the SUM over ranks and grad_scale the 1 / world of the average)."""
        self.gather_grads(zero_missing=False)
        sq = None
        for i, p in enumerate(self.params):
            if p.grad is not None:
                t = self.grad_slot(p).double().square().sum()
                sq = t if sq is None else sq + t
        return (sq.sqrt() * abs(self.grad_scale)).float() if sq is not None else torch.zeros((), device=self.device)

    def clip_grad_norm_(self, max_norm):
        """torch.nn.utils.clip_grad_norm_ for this optimizer: the clip coefficient goes into the one-shot grad_scale (no launch over the
        gradients; .grad itself is not rescaled).  Returns the norm before clipping."""
        norm = self.grad_norm()
        coef = float(max_norm) / (float(norm) + 1e-6)
        if coef < 1.0:
            self.grad_scale *= coef
        return norm

    def zero_grad(self, set_to_none=True):
        """set_to_none (default): drop the .grad references -- the next backward writes the slots afresh; no launch."""
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()
        self._handed_out = [not set_to_none and p.grad is not None for p in self.params]

    # ---- packed weights --------------------------------------------------------------------------------------------
    @staticmethod
    def _packed_shape(n, k):
        return (64 if n <= 64 else 128 if n <= 128 else 256), -(-k // 64) * 64

    def repack(self):
        """(Re)build the packed bf16 forms of every 2-D parameter [K, N]: W^T (the forward product's [N rows, K padded], N <= 256)
        and W itself (the input-gradient product's [K rows, N padded], K <= 256 -- a wider first layer has no such product)."""
        self._packed.clear()
        for i, p in enumerate(self.params):
            if p.dim() != 2 or p.shape[1] > 256:
                continue
            self._packed[i] = self._pack_pair(p)
            self._seen_version[i] = p._version
        self._table = None

    @staticmethod
    def _pack_pair(p):
        from . import dense

        with torch.no_grad():
            return (dense._pack_now(p.detach()) if p.shape[0] <= 256 else None, dense._pack_now(p.detach().t()))

    def packed_for(self, wt):
        """The packed form of `wt` (a parameter of this optimizer or its transposed view) if it is registered and current."""
        if not self._packed:
            return None
        base = wt._base if wt._base is not None else wt
        ref = getattr(base, "_dgll_flat", None)
        if ref is None or ref[0] is not self:
            return None
        i = ref[1]
        pair = self._packed.get(i)
        if pair is None:
            return None
        p = self.params[i]
        if wt.data_ptr() != p.data_ptr():
            return None
        if p._version != self._seen_version.get(i):       # somebody else wrote the parameter in place: refresh this one
            pair = self._packed[i] = self._pack_pair(p)
            self._seen_version[i] = p._version
            self._table = None
        if tuple(wt.shape) == tuple(p.shape) and wt.stride() == p.stride():
            return pair[0]
        if tuple(wt.shape) == tuple(p.shape[::-1]) and wt.stride() == p.stride()[::-1]:
            return pair[1]
        return None

    def _segments(self):
        if self._table is None:
            idx = sorted(self._packed)
            n = len(idx)
            i64, vp = C.c_int64 * max(n, 1), C.c_void_p * max(n, 1)
            begin = i64(*[self.offsets[i] for i in idx])
            end = i64(*[self.offsets[i] + self.params[i].numel() for i in idx])
            cols = (C.c_int * max(n, 1))(*[self.params[i].shape[1] for i in idx])
            pk = vp(*[self._packed[i][0].data_ptr() if self._packed[i][0] is not None else None for i in idx])
            ld = i64(*[self._packed[i][0].stride(0) if self._packed[i][0] is not None else 0 for i in idx])
            pkt = vp(*[self._packed[i][1].data_ptr() for i in idx])
            ldt = i64(*[self._packed[i][1].stride(0) for i in idx])
            self._table = (n, begin, end, cols, pk, ld, pkt, ldt)
        return self._table

    # ---- the step --------------------------------------------------------------------------------------------------
    def _runs(self, has):
        """Runs of consecutive parameters updated by one launch: same group, same (new) step count, all with a gradient."""
        runs, cur = [], None
        for i, ok in enumerate(has):
            key = (self._group_of[i], self.param_steps[i] + 1) if ok else None
            if key is None:
                cur = None
                continue
            if cur is not None and cur[0] == key:
                cur[2] = i
            else:
                cur = [key, i, i]
                runs.append(cur)
        return [(key[0], key[1], i0, i1) for key, i0, i1 in runs]

    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("FlatAdam.step takes no closure")
        if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("FlatAdam.step() cannot be captured into a HIP graph: its bias corrections are host scalars of the launch "
                               "(call it after graph.replay(), as graphs.GraphedSampledStep does)")
        has = [p.grad is not None for p in self.params]
        self.gather_grads(zero_missing=False)
        self.steps += 1
        self._handed_out = [True] * len(self.params)        # until the next zero_grad(): .grad still holds this step's gradient
        scale, self.grad_scale = float(self.grad_scale), 1.0    # one-shot
        for gi, step_no, i0, i1 in self._runs(has):
            group = self.param_groups[gi]
            begin, end = self.offsets[i0], (self.offsets[i1 + 1] if i1 + 1 < len(self.params) else self.total)
            self._update(begin, end, float(group["lr"]), (float(group["betas"][0]), float(group["betas"][1])), float(group["eps"]),
                         float(group["weight_decay"]), step_no, scale)
            for i in range(i0, i1 + 1):
                self.param_steps[i] = step_no

    def _update(self, begin, end, lr, betas, eps, weight_decay, step_no, scale):
        if self.device.type != "cuda":
            return self._update_host(begin, end, lr, betas, eps, weight_decay, step_no, scale)
        whole = begin == 0 and end == self.total
        if self.pack_weights:
            n, sb, se, cols, pk, ld, pkt, ldt = self._segments() if whole else self._segments_in(begin, end)
        else:
            n, sb, se, cols, pk, ld, pkt, ldt = (0, None, None, None, None, None, None, None)
        with _lib.on_device(self.device):
            code = _lib.lib.dgll_hip_adam_flat(
                _lib.raw_stream(self.device), self.flat.data_ptr() + 4 * begin, self.grad.data_ptr() + 4 * begin,
                self.exp_avg.data_ptr() + 4 * begin, self.exp_avg_sq.data_ptr() + 4 * begin, end - begin, lr, betas[0], betas[1], eps,
                weight_decay, step_no, scale, n, sb, se, cols, pk, ld, pkt, ldt)
        _lib.check(code, "dgll_hip_adam_flat")

    def _segments_in(self, begin, end):
        """The packed-weight table of the parameters inside [begin, end), offsets relative to `begin` (a partial update)."""
        idx = [i for i in sorted(self._packed) if begin <= self.offsets[i] and self.offsets[i] + self.params[i].numel() <= end]
        n = len(idx)
        i64, vp = C.c_int64 * max(n, 1), C.c_void_p * max(n, 1)
        return (n, i64(*[self.offsets[i] - begin for i in idx]), i64(*[self.offsets[i] + self.params[i].numel() - begin for i in idx]),
                (C.c_int * max(n, 1))(*[self.params[i].shape[1] for i in idx]),
                vp(*[self._packed[i][0].data_ptr() if self._packed[i][0] is not None else None for i in idx]),
                i64(*[self._packed[i][0].stride(0) if self._packed[i][0] is not None else 0 for i in idx]),
                vp(*[self._packed[i][1].data_ptr() for i in idx]), i64(*[self._packed[i][1].stride(0) for i in idx]))

    def _update_host(self, begin, end, lr, betas, eps, weight_decay, step_no, scale):
        """Host tensors (the gloo tests): the same arithmetic with torch ops."""
        b1, b2 = betas
        sl = slice(begin, end)
        with torch.no_grad():
            g = self.grad[sl] * scale
            if weight_decay:
                g = g + weight_decay * self.flat[sl]
            self.exp_avg[sl].lerp_(g, 1 - b1)
            self.exp_avg_sq[sl].mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1, bc2 = 1 - b1 ** step_no, 1 - b2 ** step_no
            denom = (self.exp_avg_sq[sl].sqrt() / (bc2 ** 0.5)).add_(eps)
            self.flat[sl].addcdiv_(self.exp_avg[sl], denom, value=-lr / bc1)

    def state_dict(self):
        return {"steps": self.steps, "param_steps": list(self.param_steps), "exp_avg": self.exp_avg.clone(),
                "exp_avg_sq": self.exp_avg_sq.clone(),
                "groups": [{k: g[k] for k in ("lr", "betas", "eps", "weight_decay")} for g in self.param_groups],
                "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd):
        """Moments, step counts AND hyper-parameters (a checkpoint taken at a decayed learning rate resumes at it)."""
        self.steps = int(sd["steps"])
        self.param_steps = [int(v) for v in sd.get("param_steps", [self.steps] * len(self.params))]
        if len(self.param_steps) != len(self.params):
            raise ValueError("state_dict holds %d parameters, this optimizer %d" % (len(self.param_steps), len(self.params)))
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        groups = sd.get("groups") or [{k: sd[k] for k in ("lr", "betas", "eps", "weight_decay") if k in sd}]
        if len(groups) != len(self.param_groups):
            raise ValueError("state_dict holds %d parameter groups, this optimizer %d" % (len(groups), len(self.param_groups)))
        for g, saved in zip(self.param_groups, groups):
            for k, v in saved.items():
                g[k] = tuple(v) if k == "betas" else v


def grad_slot_of(param):
    """A fresh gradient-slot view for `param` when a FlatAdam owns it, else None (the layers' backward passes call this)."""
    ref = getattr(param, "_dgll_flat", None)
    return ref[0].claim_slot(param) if ref is not None else None
