#!/usr/bin/env python3
"""fp32-output MFMA transform against a float64 product of the same bf16 operands, at the reduction lengths the kernels switch
on (resident weights up to 512, the per-chunk kernel above): the only difference allowed is fp32 accumulation order."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import dense  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
for k1, k2, n in ((256, 256, 256), (602, 0, 256), (602, 602, 256), (100, 100, 256), (256, 0, 47), (1024, 0, 128)):
    m = 8192
    x1 = torch.randn(m, k1, device=dev).to(torch.bfloat16)
    x1p = torch.zeros(m, -(-k1 // 8) * 8, device=dev, dtype=torch.bfloat16); x1p[:, :k1] = x1
    w1 = (torch.randn(n, k1, device=dev) * 0.1).to(torch.bfloat16)
    ref = x1.double() @ w1.double().t()
    a2 = wt2 = None
    if k2:
        x2 = torch.randn(m, k2, device=dev).to(torch.bfloat16)
        x2p = torch.zeros(m, -(-k2 // 8) * 8, device=dev, dtype=torch.bfloat16); x2p[:, :k2] = x2
        w2 = (torch.randn(n, k2, device=dev) * 0.1).to(torch.bfloat16)
        ref = ref + x2.double() @ w2.double().t()
        a2, wt2 = x2p[:, :k2], w2
    out = dense.transform_bf16(x1p[:, :k1], w1, a2, wt2, out_dtype=torch.float32)
    err = (out.double() - ref).abs()
    print("K = %4d + %4d, N = %3d: max |err| %.3e, rms err %.3e, rms of the product %.3e" %
          (k1, k2, n, float(err.max()), float(err.pow(2).mean().sqrt()), float(ref.pow(2).mean().sqrt())), flush=True)
