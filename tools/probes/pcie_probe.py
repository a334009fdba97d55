"""What the PCIe link gives the loading stage: (a) an asynchronous H2D copy from pinned memory, (b) the zero-copy gather of feature rows
(dgll_hip_gather_rows_mapped with every id a miss) at the Reddit row width (602 bf16 = 1204 B rows) and at a 16-byte-aligned pitch, by
grid size, (c) two such gathers on two streams at once."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dgll_amd import _lib
from dgll_amd.cache import GraphCacheServer

dev = torch.device("cuda:0")
N, F = 232_965, 602


def timed(fn, reps=8):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


src = torch.empty(64 << 20, dtype=torch.uint8, pin_memory=True)
dst = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
ms = timed(lambda: dst.copy_(src, non_blocking=True))
print("H2D copy engine, 64 MiB from pinned memory: %.3f ms = %.1f GB/s" % (ms, (64 << 20) / ms / 1e6))

for pitch in (602, 608, 640):
    store = torch.randn(N, pitch).to(torch.bfloat16).pin_memory()
    feats = store[:, :F]
    srv = GraphCacheServer(feats)
    assert srv.features.data_ptr() == feats.data_ptr()
    g = torch.Generator().manual_seed(1)
    for n in (38_000, 300_000):
        ids = torch.randint(0, N, (n,), generator=g).to(dev)
        nbytes = n * F * 2
        for bpc in (1, 2, 4, 8, 16):
            _lib.lib.dgll_hip_debug_tune(12, bpc)
            ms = timed(lambda: srv.fetch_data(ids))
            print("zero-copy gather, host pitch %d, %6d rows (%.0f MB), %2d workgroups per CU: %.3f ms = %.1f GB/s" % (pitch, n, nbytes / 1e6, bpc, ms, nbytes / ms / 1e6))
    _lib.lib.dgll_hip_debug_tune(12, 4)
    ids2 = torch.randint(0, N, (300_000,), generator=g).to(dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def both():
        srv.fetch_data(ids, stream=s1); srv.fetch_data(ids2, stream=s2)
    both(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        both()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 6 * 1e3
    print("two zero-copy gathers of 300000 rows on two streams, pitch %d: %.3f ms = %.1f GB/s together" % (pitch, ms, 2 * 300_000 * F * 2 / ms / 1e6))
    _lib.lib.dgll_hip_debug_tune(12, 0)
