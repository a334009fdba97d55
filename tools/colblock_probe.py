#!/usr/bin/env python3
"""Source-column blocking probe: the bench graph's F = 256 gather in CSR order against the same edges
walked group by group (R consecutive rows per workgroup) in ascending SOURCE order.  Upper bound of what a blocked SpMM (row
accumulators in LDS) could gain: only the gather is run.  Kill criterion: < 8 % on the 4.39 ms forward pass.

    python tools/colblock_probe.py [--rows-per-group 64] [--locality 0.9]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from dgll_amd import ops, synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def build():
    src, so = os.path.join(HERE, "probes", "colblock_gather.hip"), "/tmp/colblock_gather.so"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, src])
    lib = C.CDLL(so)
    lib.colblock_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows-per-group", default="64")
    ap.add_argument("--locality", type=float, default=0.9)
    ap.add_argument("--edges-per-group", default="", help="comma list: groups of a fixed number of EDGES instead of rows (perfect balance)")
    args = ap.parse_args()
    lib = build()
    dev = torch.device("cuda:0")
    g = synth.products_like_graph(dev, seed=0, locality=args.locality, exact=True, permute_ids=True)
    g, _ = g.reorder(seed=0)
    n, nnz = g.n_rows, g.nnz
    x = ops.alloc_features(n, 256, torch.bfloat16, dev)
    x.copy_(torch.randn(n, 256, device=dev).to(torch.bfloat16))
    table = x.contiguous()
    g.plan()
    for _ in range(2):
        ops.spmm_raw(g, x, reduce="mean")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        ops.spmm_raw(g, x, reduce="mean")
    b.record()
    torch.cuda.synchronize()
    print("bench graph (locality %.2f): %d rows, %d edges; shipped spmm_csr_kernel F=256 mean: %.3f ms" % (args.locality, n, nnz, a.elapsed_time(b) / 5))
    row = g.row_index()
    stream = torch.cuda.current_stream().cuda_stream

    def run(idx, ptr, n_groups, rpg, remap, u, label):
        out = torch.empty(n_groups * rpg * 256, dtype=torch.bfloat16, device=dev)
        for _ in range(2):
            lib.colblock_gather(stream, table.data_ptr(), idx.data_ptr(), ptr.data_ptr(), n_groups, rpg, out.data_ptr(), remap, u)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            lib.colblock_gather(stream, table.data_ptr(), idx.data_ptr(), ptr.data_ptr(), n_groups, rpg, out.data_ptr(), remap, u)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        print("   R=%4d %-34s remap=%d U=%d: %.3f ms  (%.2f TB/s gathered)" % (rpg, label, remap, u, ms, nnz * 512 / ms / 1e9))
        return ms

    for rpg in [int(v) for v in args.rows_per_group.split(",")]:
        n_groups = (n + rpg - 1) // rpg
        bounds = torch.arange(0, n_groups + 1, device=dev, dtype=torch.int64).mul_(rpg).clamp_(max=n)
        ptr = g.rowptr[bounds].contiguous()
        col = g.col
        # (1) CSR order: the group's rows one after the other
        for remap in (0, 1):
            run(col, ptr, n_groups, rpg, remap, 4, "CSR order (row after row)")
        # (2) per group, ascending source id
        key = (row // rpg) * n + col.long()
        order = torch.argsort(key)
        sorted_col = col[order].contiguous()
        del key, order
        for remap in (0, 1):
            for u in (4, 8):
                run(sorted_col, ptr, n_groups, rpg, remap, u, "group's edges by ascending source")
        # (3) the same with duplicates of a source inside a group gathered once (what an LDS-accumulating kernel could NOT do:
        #     every edge has its own destination row; listed only as the bound of 'one fetch per distinct source per group')
        del sorted_col
        torch.cuda.empty_cache()
    # groups of equal EDGE counts (cut anywhere, also inside a row: the long-row chunking of the shipped kernel taken to its limit):
    # the row-group form above is dominated by the hub groups' tails, this one isolates what the source order does to the gather
    for epg in [int(v) for v in args.edges_per_group.split(",") if v]:
        n_groups = (nnz + epg - 1) // epg
        ptr = torch.arange(0, n_groups + 1, device=dev, dtype=torch.int64).mul_(epg).clamp_(max=nnz)
        rpg = max(1, epg // 50)
        for remap in (0, 1):
            run(g.col, ptr, n_groups, rpg, remap, 4, "E=%d per group, CSR order" % epg)
        key = (torch.arange(nnz, device=dev) // epg) * n + g.col.long()
        sorted_col = g.col[torch.argsort(key)].contiguous()
        del key
        for remap in (0, 1):
            run(sorted_col, ptr, n_groups, rpg, remap, 4, "E=%d per group, ascending source" % epg)
        del sorted_col
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
