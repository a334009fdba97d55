#!/usr/bin/env python3
"""Which tensor OPS (not kernels) of a sampled GraphSAGE step still run as torch kernels: one eager forward + loss + backward at
the Reddit shape's upper bounds under torch.profiler, grouped by operator."""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import nn as dnn, ops  # noqa: E402
from dgll_amd.graphs import PaddedBlock  # noqa: E402
from dgll_amd.optim import FlatAdam  # noqa: E402

dev = torch.device("cuda:0")
fanouts, batch, feats, classes = [25, 10, 10], 1024, 602, 41
order = list(reversed(fanouts))
rows = [batch, batch * order[0], batch * order[0] * order[1]]
rows = [rows[0], int(rows[1] * 0.96), int(rows[2] * 0.95)]
total = sum(rows)
store = ops.alloc_features(total, feats, torch.bfloat16, dev); store.normal_()
offs = [0, rows[0], rows[0] + rows[1], total]
features = [store[offs[h]:offs[h + 1]] for h in range(3)]
agg_all = ops.alloc_features(total, feats, torch.bfloat16, dev); agg_all.normal_()
reduced = agg_all[offs[2]:offs[3]]; reduced._dgll_stack = agg_all
blocks = [PaddedBlock.make(rows[h], order[h], dev, cols=rows[h + 1]) for h in range(2)] + [None]
labels = torch.randint(0, classes, (rows[0],), device=dev)
model = dnn.GraphSage(feats, [256, 256, classes], fanouts).to(dev)
opt = FlatAdam(list(model.parameters()), lr=1e-3)


def step():
    opt.zero_grad(set_to_none=True)
    out = model.forward_sampled(features, blocks, last_hop_reduced=reduced)
    loss = ops.cross_entropy(out, labels)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
    for _ in range(4):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=70))
if len(sys.argv) > 1 and sys.argv[1] == "shapes":      # every torch op that launches a kernel, with shapes and the dgll_amd line that issued it
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode

    VIEWS = ("view", "select", "slice", "as_strided", "expand", "reshape", "t.default", "transpose", "unsqueeze", "squeeze", "detach", "alias",
             "empty", "_unsafe_view", "permute", "narrow", "unbind", "split", "is_", "sym_", "stride", "size", "numel", "_local_scalar")

    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            if not any(v in name for v in VIEWS):
                frames = [f for f in traceback.extract_stack() if "dgll_amd" in f.filename][-2:]
                shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)]
                print("%-34s %-44s %s" % (name.replace("aten.", ""), str(shapes)[:44],
                                          " <- ".join("%s:%d %s" % (os.path.basename(f.filename), f.lineno, f.name) for f in reversed(frames))))
            return func(*args, **(kwargs or {}))

    with Log():
        step()
    torch.cuda.synchronize()
