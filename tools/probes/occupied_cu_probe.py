#!/usr/bin/env python3
"""How the persistent whole-CU kernels (resident-weights MFMA transform, split-K weight gradient: one workgroup per CU with the
CU's whole register file / 128 KiB of LDS, rows statically split over the workgroups) behave when a few CUs are held by somebody
else's wavefronts -- what a collective's channel kernels on the communication stream do to them in a multi-GPU step.  Occupiers:
n single-wavefront spin kernels on n side streams (torch.cuda._sleep), each pinning one wavefront slot of some CU for ~3 ms.
    python tools/probes/occupied_cu_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from dgll_amd import _lib, dense, ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    m = 306_000                                        # one rank's rows at 8 ranks of the products-sized graph
    h = ops.alloc_features(m, 256, torch.bfloat16, dev); h.normal_()
    a = ops.alloc_features(m, 256, torch.bfloat16, dev); a.normal_()
    g = ops.alloc_features(m, 256, torch.bfloat16, dev); g.normal_()
    w1, w2 = torch.randn(256, 256, device=dev) / 16, torch.randn(256, 256, device=dev) / 16
    side = [torch.cuda.Stream(dev) for _ in range(32)]

    def timed(fn, occupiers):
        best = []
        for _ in range(5):
            torch.cuda.synchronize()
            for s in side[:occupiers]:
                with torch.cuda.stream(s):
                    torch.cuda._sleep(6_000_000)                # ~3 ms
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1))
        return sorted(best)[len(best) // 2]

    cases = [("transform 256+256 -> 256 (resident weights)", lambda: dense.transform_bf16(h, w1, a, w2, relu=True)),
             ("weight gradients (h, agg)^T . g (split-K)", lambda: dense.grad_weight_pair(h, a, g)),
             ("dual g . W^T x 2", lambda: dense.transform_bf16_dual(g, w1, w2))]
    for kernel in ("default", "4-wave"):
        _lib.check(_lib.lib.dgll_hip_debug_tune(4, 1 if kernel == "4-wave" else 0), "tune")
        for name, fn in cases:
            fn()
            row = ["%7.1f us" % (timed(fn, k) * 1e3) for k in (0, 4, 16, 32)]
            print("%-8s %-46s occupiers 0 / 4 / 16 / 32: %s" % (kernel, name, "  ".join(row)), flush=True)
    _lib.check(_lib.lib.dgll_hip_debug_tune(4, 0), "tune")


if __name__ == "__main__":
    main()
