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


class FlatAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, pack_weights=True):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdam got no parameter that requires grad")
        dev = self.params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise ValueError("FlatAdam takes fp32 parameters on one device")
        self.device = dev
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.steps = 0
        self.grad_scale = 1.0          # RaCoM sets 1 / world_size (the reference's average, MQGCN.py:64) instead of a div_ launch
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

    def gather_grads(self):
        """Make the gradient buffer hold every parameter's gradient: slots written in place are left alone, a gradient that autograd
        produced elsewhere is copied in, a parameter without gradient gets zeros.  Afterwards every .grad IS its slot."""
        for i, p in enumerate(self.params):
            g = p.grad
            if g is not None and g.data_ptr() == self._slot_ptr(i) and g.is_contiguous():
                continue
            slot = self.grad_slot(p)
            if g is None:
                slot.zero_()
            else:
                slot.copy_(g)
            p.grad = slot

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
    def step(self):
        self.gather_grads()
        self.steps += 1
        self._handed_out = [True] * len(self.params)        # until the next zero_grad(): .grad still holds this step's gradient
        if self.device.type != "cuda":
            return self._step_host()
        n, begin, end, cols, pk, ld, pkt, ldt = self._segments() if self.pack_weights else (0, None, None, None, None, None, None, None)
        with _lib.on_device(self.device):
            code = _lib.lib.dgll_hip_adam_flat(
                _lib.raw_stream(self.device), self.flat.data_ptr(), self.grad.data_ptr(),
                self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.total, self.lr, self.betas[0], self.betas[1], self.eps,
                self.weight_decay, self.steps, float(self.grad_scale), n, begin, end, cols, pk, ld, pkt, ldt)
        _lib.check(code, "dgll_hip_adam_flat")

    def _step_host(self):
        """Host tensors (the gloo tests): the same arithmetic with torch ops."""
        b1, b2 = self.betas
        with torch.no_grad():
            g = self.grad * self.grad_scale
            if self.weight_decay:
                g = g + self.weight_decay * self.flat
            self.exp_avg.lerp_(g, 1 - b1)
            self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1, bc2 = 1 - b1 ** self.steps, 1 - b2 ** self.steps
            denom = (self.exp_avg_sq.sqrt() / (bc2 ** 0.5)).add_(self.eps)
            self.flat.addcdiv_(self.exp_avg, denom, value=-self.lr / bc1)

    def state_dict(self):
        return {"steps": self.steps, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd):
        self.steps = int(sd["steps"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])


def grad_slot_of(param):
    """A fresh gradient-slot view for `param` when a FlatAdam owns it, else None (the layers' backward passes call this)."""
    ref = getattr(param, "_dgll_flat", None)
    return ref[0].claim_slot(param) if ref is not None else None
