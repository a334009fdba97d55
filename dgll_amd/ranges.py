"""Named profiler ranges on the pipeline stages -- the reference's only observability surface:
`torch.autograd.profiler.record_function('gpu-load' / 'gpu-compute')` around the two halves of a step (FeatureCache/gs.py:88,93) and
'cache-idxload' / 'cache-index' / 'cache-gpu' / 'cache-cpu' inside GraphCacheServer.fetch_data (storage.py:164-195), printed as
`prof.key_averages().table(...)` (gs.py:113).

Here the same names (plus the stages the reference does not have) behind ONE switch, DGLL_PROFILE_RANGES=1 (or ranges.enable()):

    sample            drawing a batch on the host (sampling thread / the K sampler workers)
    gpu-load          the loading stage's work for one batch (id upload, cache gather, outermost-hop reduction, CSR blocks)
    cache-index       fetch_data / aggregate_data: id upload, output allocation, snapshot of the (slot map, cache block) pair
    cache-gpu         the one launch that serves hits from the HBM cache AND misses from the pinned host rows
    cache-cpu         get_feat_from_server: rows gathered on the host (cache fill / refresh, the no-cache path)
    consume           the training step on a loaded batch (the reference's 'gpu-compute')
    exchange          a halo exchange (start + wait) of the partitioned path
    racom-allreduce   the gradient bucket's all-reduce (launch + wait)

Off (the default) `rng(name)` returns one shared no-op context manager: no allocation, no profiler call.  On, it is
torch.profiler.record_function: the ranges show up in torch.profiler / rocprofv3 --marker-trace timelines and in
tools/range_table.py's per-range table."""
import os

import torch

_ON = os.environ.get("DGLL_PROFILE_RANGES", "0") not in ("", "0")

NAMES = ("sample", "gpu-load", "cache-index", "cache-gpu", "cache-cpu", "consume", "exchange", "racom-allreduce")


class _Null:
    __slots__ = ()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL = _Null()


def enable(on=True):
    global _ON
    _ON = bool(on)


def enabled():
    return _ON


def rng(name):
    """Context manager: a named range when the switch is on, a shared no-op otherwise."""
    return torch.profiler.record_function(name) if _ON else _NULL


def profile(**kw):
    """torch.profiler.profile that also records the ranges of the PIPELINE'S OWN THREADS (sampler workers, the loading stage): a
    plain torch.profiler.profile only sees the thread that started it -- the reference's loop is single-threaded
    (FeatureCache/gs.py:82), this pipeline is not.  Keyword arguments as torch.profiler.profile."""
    from torch._C._profiler import _ExperimentalConfig

    kw.setdefault("experimental_config", _ExperimentalConfig(profile_all_threads=True))
    return torch.profiler.profile(**kw)
