#!/usr/bin/env python3
"""A/B of the flattened SpMM kernel (spmm_csr_flat_kernel, dgll_hip_debug_tune(13, v): 1 = off, 2 = whenever it applies) against the
wave-per-row kernel on the bench graph: the launch kinds of the headline step, interleaved, + agreement of the results."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dgll_amd  # noqa: E402,F401
from dgll_amd import _lib, ops, synth  # noqa: E402


def tune(k, v):
    _lib.check(_lib.lib.dgll_hip_debug_tune(k, v), "tune")


def main():
    dev = torch.device("cuda:0")
    locality = float(sys.argv[1]) if len(sys.argv) > 1 else 0.9
    edges = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["512"])]
    raw = synth.products_like_graph(dev, seed=0, locality=locality, exact=True, permute_ids=True)
    raw, _ = raw.reorder(seed=0)
    n = raw.n_rows
    for epw in edges:
        tune(14, epw)
        g = dgll_amd.CSRGraph(raw.rowptr, raw.col, None, n, n, check=False)     # a fresh plan with this schedule
        gt = g.transpose()[0]
        gt.plan()
        scale = gt.mean_scale_transposed() if hasattr(gt, "mean_scale_transposed") else None
        val = torch.rand(gt.nnz, device=dev)
        cases = []
        for feat, dt in ((256, torch.bfloat16), (100, torch.bfloat16), (128, torch.bfloat16), (100, torch.float32)):
            x = ops.alloc_features(n, feat, dt, dev, pad_to=64 if dt == torch.bfloat16 else 4)
            x.copy_(torch.randn(n, feat, device=dev).to(dt))
            cases.append(("F=%d %s forward mean" % (feat, str(dt)[6:]), lambda x=x: ops.spmm_raw(g, x, reduce="mean")))
            if feat == 256:
                gate = ops.alloc_features(n, feat, dt, dev)
                gate.copy_(torch.randn(n, feat, device=dev).to(dt))
                out0 = ops.alloc_features(n, feat, dt, dev)
                out0.copy_(torch.randn(n, feat, device=dev).to(dt))

                def bwd(x=x, gate=gate, out0=out0):
                    out = out0.clone()
                    return ops.spmm_raw(gt, x, val=val, reduce="sum", out=out, accumulate=1, gate=gate)

                cases.append(("F=256 bf16 transposed weighted accumulate + gate", bwd))
        for name, fn in cases:
            res, ms = {}, {}
            for mode in (1, 2, 1, 2):
                tune(13, mode)
                for _ in range(2):
                    y = fn()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                a.record()
                for _ in range(5):
                    y = fn()
                b.record()
                torch.cuda.synchronize()
                ms.setdefault(mode, []).append(a.elapsed_time(b) / 5)
                res[mode] = y.float().clone()
                if mode == 2:
                    assert torch.equal(fn().float(), res[2]), "flat kernel is not bit-reproducible"
            diff = float((res[2] - res[1]).abs().max())
            ref = float(res[1].abs().max())
            print("E=%4d %-52s wave-per-row %s ms | flattened %s ms | max |diff| %.3e (max |y| %.2f)" % (
                epw, name, " / ".join("%.3f" % t for t in ms[1]), " / ".join("%.3f" % t for t in ms[2]), diff, ref))
    tune(13, 0)
    tune(14, 512)


if __name__ == "__main__":
    main()
