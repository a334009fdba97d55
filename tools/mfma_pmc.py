#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes over the MFMA transform kernel (fused SAGE transform, K = 256 + 256, N = 256)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import _lib, dense, ops  # noqa: E402

dev = torch.device("cuda:0")
M, K, N = 2_449_029, 256, 256
h = ops.alloc_features(M, K, torch.bfloat16, dev); h.copy_(torch.randn(M, K, device=dev))
agg = ops.alloc_features(M, K, torch.bfloat16, dev); agg.copy_(torch.randn(M, K, device=dev))
ws = torch.randn(K, N, device=dev).to(torch.bfloat16).t().contiguous()
wn = torch.randn(K, N, device=dev).to(torch.bfloat16).t().contiguous()
for variant in (0, 1):
    _lib.lib.dgll_hip_debug_tune(4, variant)
    for _ in range(3):
        dense.transform_bf16(h, ws, agg, wn, relu=True)
    torch.cuda.synchronize()
