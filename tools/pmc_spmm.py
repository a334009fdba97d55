#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes (HBM traffic of the dominant kernel).

Launches, in this order and each `--reps` times:
  1. calibration: dgll_hip_spmm_csr on the identity-gather structure (row i reads row i): a pure streaming read of
     N*F*2 bytes + the same written, in the kernel's own access pattern (16 B per lane) -- the known byte count the
     guide asks FETCH_SIZE/WRITE_SIZE to be calibrated against;
  2. the headline launch: mean-SpMM, F=256 bf16, products-shaped graph.
Run it once per counter group (FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum), kernel-trace only.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dgll_amd  # noqa: E402
from dgll_amd import ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--feat", type=int, default=256)
ap.add_argument("--locality", type=float, default=None)
args = ap.parse_args()
dev = torch.device("cuda:0")
g = synth.products_like_graph(dev, seed=0) if args.locality is None else synth.products_like_graph(dev, seed=0, locality=args.locality)
n = g.n_rows
x = torch.randn(n, args.feat, device=dev).to(torch.bfloat16)
ident = dgll_amd.CSRGraph.fixed_fanout(n, 1, dev)
ident.plan(); g.plan()
torch.cuda.synchronize()
for _ in range(args.reps):
    ops.spmm_raw(ident, x, reduce="sum")
torch.cuda.synchronize()
for _ in range(args.reps):
    ops.spmm_raw(g, x, reduce="mean")
torch.cuda.synchronize()
print("calibration bytes read=%d written=%d ; headline nnz=%d n=%d" % (n * args.feat * 2 + n * 12, n * args.feat * 2, g.nnz, n))
