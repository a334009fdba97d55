"""MQ-GNN style mini-batch pipeline: bounded queues between a sampling/feature-loading producer and the training
consumer, each stage on its own HIP stream.

Reference: /root/reference/dgll/GPU Accelerator/buffer_queues.py:22-46 (`sample_generator`: iterate the dataloader,
stage the batch on `d_stream`, `gpu_queue.put`), :74-113 (`sample_consumer`: `gpu_queue.get`, forward/backward on
`c_stream`, gradient sharing on `g_stream`), MQGCN.py:98,150-155 (BUFFER_SIZE = 4, one producer + one consumer thread,
`Queue(maxsize=4)` + `Condition`), README.md:27-29 (three queues: sampling -> feature loading -> training).

Here, the three queues of README.md:27-29 as two producer stages and the consumer: a SAMPLING thread draws the batches on
the host (stdlib RNG stream, bit-exact with the reference -- one thread, so the stream stays sequential) into a bounded
queue; a LOADING thread takes them, fetches the input features through the GraphCacheServer on a side stream (hot rows
from HBM, misses over PCIe from pinned memory), records an event and fills the second bounded queue; the consumer waits
on the event from its compute stream.  Sampling of batch i+2, loading of batch i+1 and training on batch i overlap.  The end of the epoch is signalled with a sentinel
(buffer_queues.py:43-46 sets a flag under the Condition).
"""
import os
import threading
import time

import numpy as np
import torch

from .ranges import rng

# The in-place native loads fetch the outermost hop's uncached rows into HBM ahead of its reduction (csrc/gather.hip: list_misses_kernel,
# stage_rows_kernel); 0 = the reduction reads them zero-copy itself.  DGLL_LOADER_STAGE_BLOCKS: workgroups of the fetch.
STAGE_MISSES = os.environ.get("DGLL_LOADER_STAGE_MISSES", "1") != "0"
STAGE_BLOCKS = int(os.environ.get("DGLL_LOADER_STAGE_BLOCKS", "0"))       # 0 = the library's 20
UPLOAD_BLOCKS_ALONE = int(os.environ.get("DGLL_LOADER_UPLOAD_BLOCKS_ALONE", "128"))
STAGE_CAP = int(os.environ.get("DGLL_LOADER_STAGE_CAP", "0"))        # rows of the staging buffer (0 = every uncached node fits); tests

_DONE = object()


class AdaptiveQueue:
    """Bounded FIFO whose bound moves with the balance between its producer and its consumer -- README.md:29's "adaptive
    queue-sizing strategy to balance computation and memory efficiency" (the reference's code keeps BUFFER_SIZE = 4,
    MQGCN.py:98).  Every `window` items the queue looks at how long the consumer sat starved (get() found it empty) and how
    long the producer sat blocked (put() found it full):
      * the consumer starved for more than 5 % of the window  -> double the bound (up to `max_size`): more run-ahead hides
        the producer's jitter (a hub-heavy batch takes several times the median to sample and load);
      * the consumer never starved and the producer was blocked for more than half of the window -> shrink by one (down to
        `min_size`): the depth beyond what keeps the consumer busy is memory held for nothing.
    `max_size` is a MEMORY bound: MiniBatchPipeline sets it from the bytes of a loaded batch against a budget of device
    memory (set_memory_bound)."""

    def __init__(self, size=4, min_size=2, max_size=64, window=8, adaptive=True):
        self.size, self.min_size, self.max_size, self.window, self.adaptive = int(size), int(min_size), int(max_size), int(window), adaptive
        self._items = []
        self._cv = threading.Condition()
        self._starved = self._blocked = 0.0
        self._t_window = time.perf_counter()
        self._count = 0
        self.history = [int(size)]          # the bound after every adjustment (diagnostics / tests)

    def set_memory_bound(self, batch_bytes, budget_bytes):
        with self._cv:
            if batch_bytes > 0:
                self.max_size = max(self.min_size, min(self.max_size, int(budget_bytes // batch_bytes)))
                if self.size > self.max_size:
                    self.size = self.max_size
                    self.history.append(self.size)

    def put(self, item):
        with self._cv:
            if len(self._items) >= self.size:
                t0 = time.perf_counter()
                while len(self._items) >= self.size:
                    self._cv.wait()
                self._blocked += time.perf_counter() - t0
            self._items.append(item)
            self._cv.notify_all()

    def get(self):
        with self._cv:
            if not self._items:
                t0 = time.perf_counter()
                while not self._items:
                    self._cv.wait()
                self._starved += time.perf_counter() - t0
            item = self._items.pop(0)
            self._count += 1
            if self.adaptive and self._count % self.window == 0:
                self._adapt()
            self._cv.notify_all()
            return item

    def _adapt(self):
        now = time.perf_counter()
        span = max(now - self._t_window, 1e-9)
        starved, blocked = self._starved / span, self._blocked / span
        if starved > 0.05 and self.size < self.max_size:
            self.size = min(self.size * 2, self.max_size)
            self.history.append(self.size)
        elif starved == 0.0 and blocked > 0.5 and self.size > self.min_size:
            self.size -= 1
            self.history.append(self.size)
        self._starved = self._blocked = 0.0
        self._t_window = now

    def qsize(self):
        with self._cv:
            return len(self._items)


class OrderedHandoff:
    """Hand-over between K producers that each own whole batches and ONE consumer that must see the batches in order: put(i, item)
    blocks while i is more than `capacity` ahead of the next batch to be consumed; get() returns batch 0, 1, 2, ... (None after
    close(n) once n batches were taken).  The reorder buffer of the multi-threaded sampling stage."""

    def __init__(self, capacity):
        self.capacity = int(capacity)
        self._slots, self._next, self._total = {}, 0, None
        self._cv = threading.Condition()

    def put(self, index, item):
        """False once the hand-over was closed (the consumer is gone or another producer failed): the caller stops producing."""
        with self._cv:
            while index >= self._next + self.capacity and self._total is None:
                self._cv.wait()
            if self._total is not None and index >= self._total:
                return False
            self._slots[index] = item
            self._cv.notify_all()
            return True

    @property
    def closed(self):
        return self._total is not None

    def close(self, total):
        with self._cv:
            self._total = int(total) if self._total is None else min(self._total, int(total))
            self._cv.notify_all()

    def get(self):
        with self._cv:
            while self._next not in self._slots:
                if self._total is not None and self._next >= self._total:
                    return None
                self._cv.wait()
            item = self._slots.pop(self._next)
            self._next += 1
            self._cv.notify_all()
            return item


class PipelineStopped(Exception):
    """Raised inside a producer thread that waited for a buffer while the pipeline was being stopped (never reaches the user)."""


class PinnedRing:
    """A few page-locked int64 host buffers handed round: the outermost hop's positions / ids are written into one by the native
    sampler and uploaded from it by an asynchronous DMA (a pageable source is staged by the runtime: the loading thread sat in that
    copy for 2-3 ms per batch).  A buffer returns with the event that marks its upload complete and is reused after it.
    `stop` (threading.Event, set by the pipeline when a stage failed or the consumer left): a thread waiting for a free buffer
    gives up with PipelineStopped instead of waiting for a release that will never come."""

    def __init__(self, n_buffers, capacity, dtype=torch.int64, stop=None):
        import queue

        self._empty = queue.Empty
        self.stop = stop
        self._free = queue.Queue()
        self.buffers = [torch.empty(int(capacity), dtype=dtype, pin_memory=True) for _ in range(int(n_buffers))]
        for i in range(len(self.buffers)):
            self._free.put((i, None))
        self.capacity = int(capacity)

    def acquire(self, n):
        if n > self.capacity:
            return None
        while True:
            try:
                i, event = self._free.get(timeout=0.05)
                break
            except self._empty:
                if self.stop is not None and self.stop.is_set():
                    raise PipelineStopped() from None
        if event is not None:
            event.synchronize()
        return self.buffers[i].numpy()[:n], i

    def release(self, token, event=None):
        if token is not None:
            self._free.put((token, event))


class _Positions:
    """Stand-in for the outermost hop's int64 position array when the native sampler pool handed it over already narrowed (only its
    length is ever asked for: the loading stage uploads the narrowed copy)."""

    def __init__(self, n):
        self.shape = (int(n),)

    def numel(self):
        return self.shape[0]


class SamplerPool:
    """K NATIVE sampler threads behind one in-order hand-over (csrc/sampler.hip: dgll_host_sampler_pool_*): every batch is drawn under
    its own seed -- random.seed(batch_seed(base_seed, epoch, i)): the ids of the reference's loop, of oracle/sampler.py and of the
    Python-thread path, bit for bit -- straight into one of `n_slots` pinned slot buffers in the loading stage's upload layout.
    Python only dequeues (`next()`: blocks without the interpreter lock) and gives slots back when their uploads have completed.
    Replaces MiniBatchPipeline's K Python sampler threads, whose per-batch bookkeeping contended for the interpreter lock with the
    loading and the consuming thread (more than 8 of them made the epoch slower)."""

    HOLD = 8        # delivered batches whose slots may be waiting for their upload (queue of 2, the loader's batch, uploads in flight)

    def __init__(self, indptr, indices, train_nodes, batch_size, fanouts, base_seed, epoch, n_threads, max_degree):
        import ctypes as C

        from . import _lib
        from .sampling.fast_sampler import FastNeighborSampler, _setsize

        self._lib, self._C = _lib, C
        order = [int(f) for f in reversed(fanouts)]              # sampling order (base_sampler.py:30-58)
        L = self.L = len(order)
        self.batch_size, self.order = int(batch_size), order
        rows = [self.batch_size]
        for f in order[:-1]:
            rows.append(rows[-1] * f)
        self.rows_cap = rows
        caps = [rows[h] * order[h] for h in range(L)]
        self.cap_outer = caps[-1]
        off, o = {"seeds": 0, "src": [], "ptr": []}, rows[0]      # the layout of FastNeighborSampler.sample_seeded(staging=)
        for h in range(L - 1):
            off["src"].append(o)
            o += caps[h]
        for h in range(L):
            off["ptr"].append(o)
            o += rows[h] + 1
        self.offsets, self.entries = off, o
        assert o == FastNeighborSampler.staging_entries(batch_size, fanouts)
        self.pos_dtype = torch.int16 if max_degree < (1 << 15) else (torch.int32 if max_degree < (1 << 31) else torch.int64)
        self.n_threads = int(n_threads)
        self.n_slots = self.n_threads + self.HOLD + 2
        # page-locked slots: the loading stage uploads them with a kernel that reads host memory (DGLL_SAMPLER_POOL_PINNED=0, tests:
        # pageable slots -- dgll_hip_load_sampled_batch detects them and takes the copy engine)
        pin = torch.cuda.is_available() and os.environ.get("DGLL_SAMPLER_POOL_PINNED", "1") != "0"
        self.staged = [torch.empty(self.entries, dtype=torch.int64, pin_memory=pin) for _ in range(self.n_slots)]
        self.pos = [torch.empty(self.cap_outer, dtype=self.pos_dtype, pin_memory=pin) for _ in range(self.n_slots)]
        self._keep = (np.ascontiguousarray(indptr, dtype=np.int64), np.ascontiguousarray(indices, dtype=np.int64),
                      np.ascontiguousarray(train_nodes.numpy() if isinstance(train_nodes, torch.Tensor) else train_nodes, dtype=np.int64))
        self.train = torch.from_numpy(self._keep[2])
        fan = np.array(order, dtype=np.int64)
        sets = np.array([_setsize(f) for f in order], dtype=np.int64)
        off_src = np.array(off["src"] + [0], dtype=np.int64)
        off_ptr = np.array(off["ptr"], dtype=np.int64)
        sb = (C.c_void_p * self.n_slots)(*[t.data_ptr() for t in self.staged])
        pb = (C.c_void_p * self.n_slots)(*[t.data_ptr() for t in self.pos])
        handle = C.c_void_p()
        _lib.check(_lib.lib.dgll_host_sampler_pool_create(
            C.byref(handle), self._keep[0].ctypes.data, self._keep[1].ctypes.data, self._keep[2].ctypes.data, len(self._keep[2]),
            self.batch_size, fan.ctypes.data, sets.ctypes.data, L, int(base_seed), int(epoch), self.n_threads, self.n_slots, sb,
            self.entries, off["seeds"], off_src.ctypes.data, off_ptr.ctypes.data, pb, self.pos[0].element_size()),
            "dgll_host_sampler_pool_create")
        self._handle = handle
        self._lock = threading.Lock()
        self._pending = []                 # (slot, [events]) of delivered batches whose uploads were issued
        self._parts = {}                   # slot -> events released so far (a slot has two buffers: staged and positions)
        self.delivered = self.released = 0
        self.sample_ms = 0.0

    def next(self):
        """(batch index, slot, rows per hop, edges per hop) of the next batch in order, or None at the end of the epoch."""
        C = self._C
        out = (C.c_int64 * (3 + 2 * 8))()
        ms = C.c_double(0.0)
        self.reap(block=self.delivered - self.released >= self.HOLD)
        code = self._lib.lib.dgll_host_sampler_pool_next(self._handle, out, C.byref(ms))
        if code == 1:
            return None
        self._lib.check(code, "dgll_host_sampler_pool_next")
        L = int(out[2])
        self.delivered += 1
        self.sample_ms += ms.value
        return int(out[0]), int(out[1]), [int(out[3 + h]) for h in range(L)], [int(out[3 + L + h]) for h in range(L)]

    def release_part(self, slot, event):
        """One of the slot's two buffers (staged arrays, positions) was uploaded behind `event` (None: it will not be uploaded)."""
        with self._lock:
            got = self._parts.setdefault(slot, [])
            got.append(event)
            if len(got) == 2:
                self._pending.append((slot, [e for e in got if e is not None]))
                del self._parts[slot]

    def reap(self, block=False):
        """Give the slots whose uploads have completed back to the workers; block: wait until at least one came back."""
        while True:
            with self._lock:
                keep, done = [], []
                for slot, events in self._pending:
                    (done if all(e.query() for e in events) else keep).append((slot, events))
                self._pending = keep
                first = keep[0] if keep else None
            for slot, _ in done:
                self._lib.check(self._lib.lib.dgll_host_sampler_pool_release(self._handle, slot), "dgll_host_sampler_pool_release")
                self.released += 1
            if done or not block:
                return
            if first is not None:
                for e in first[1]:
                    e.synchronize()
            else:
                time.sleep(0.0002)           # the loading stage has not issued the oldest batch's upload yet

    def close(self):
        if self._handle is not None:
            self._lib.lib.dgll_host_sampler_pool_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class _PoolPart:
    """What the loading stage calls a ring (release(token, event)) for one of the two pinned buffers of a pool slot."""

    def __init__(self, pool):
        self.pool = pool

    def release(self, token, event=None):
        if token is not None:
            self.pool.release_part(token, event)


class Batch:
    """One loaded mini-batch.  CONTRACT for batches written in place (static_set is not None, MiniBatchPipeline.use_static_sets): features,
    last_hop_reduced, labels, blocks, input_nodes and the subgraphs' _src / indptr are VIEWS of the captured step's input set (and of
    the loading stage's per-set scratch), not copies.  They are valid from `ready` until the step that consumes the batch has been
    issued -- GraphedSampledStep.__call__ / .eager record the set's `free` event, and the loading stage rewrites the set behind it,
    a few batches later.  A consumer that wants to read them afterwards (evaluation, logging, record_access) must copy what it needs
    BEFORE calling the step, on the stream that waited for `ready`; a consumer that never calls the step on such a batch must record
    `step.sets[b.static_set].free` itself (GraphedSampledStep._mark_free) or the set is rewritten without waiting for its reads."""
    __slots__ = ("input_nodes", "output_nodes", "subgraphs", "features", "labels", "ready", "step", "last_hop_reduced", "blocks",
                 "static_set")

    def __init__(self):
        self.input_nodes = self.output_nodes = self.subgraphs = self.features = self.labels = self.ready = None
        self.last_hop_reduced = None
        self.static_set = None      # use_static_sets: index of the GraphedSampledStep input set this batch was written into
        self.blocks = None          # build_blocks: [CSR block of hop 0, hop 1, ...] on the device (None where the hop needs none)
        self.step = 0


class MiniBatchPipeline:
    def __init__(self, dataloader, cache=None, labels=None, queue_size=4, device="cuda", hops=None, memory_fraction=0.1,
                 record_access=False, reduce_last_hop=None, sampler_threads=0, base_seed=0, epoch=0, device_graph=None,
                 build_blocks=False):
        """dataloader: dgll_amd.dataloader.DataLoader; cache: GraphCacheServer (None: features come from
        dataloader.Dgraph.get_features on the host and are copied); hops: optional callable batch -> list of id tensors
        whose features are needed (default: the input nodes only, graphage.py:52).
        queue_size: an int fixes the bound of the loaded-batch queue (MQGCN.py:98 uses 4); "auto" starts at 4 and lets it
        adapt (AdaptiveQueue), never holding more loaded batches than fit in `memory_fraction` of the free device memory.
        record_access: feed the cache's access counters (GraphCacheServer.refresh_from_access).
        reduce_last_hop: "mean" / "sum" (needs `cache` and `hops`): the outermost hop's features are not fetched; the batch
        carries their per-destination reduction over the outermost sampled block instead (Batch.last_hop_reduced, features[-1] is
        None) -- GraphCacheServer.aggregate_data reads the cache / the pinned host rows directly, and a queued batch is an order of
        magnitude smaller.  For GraphSage.forward_sampled(..., last_hop_reduced=...)."""
        self.dataloader, self.cache, self.labels = dataloader, cache, labels
        self.device = torch.device(device)
        adaptive = queue_size == "auto"
        self.queue = AdaptiveQueue(4 if adaptive else int(queue_size), adaptive=adaptive)      # loaded batches
        self.sampled = AdaptiveQueue(2, adaptive=False)           # sampled, not yet loaded: a short hand-over queue
        self.memory_fraction, self.record_access = memory_fraction, record_access
        if reduce_last_hop is not None and (cache is None or hops is None or reduce_last_hop not in ("mean", "sum")):
            raise ValueError("reduce_last_hop ('mean' / 'sum') needs a GraphCacheServer and a `hops` callback")
        self.reduce_last_hop = reduce_last_hop
        self.sampler_threads, self.base_seed, self.epoch = int(sampler_threads), int(base_seed), int(epoch)
        if self.sampler_threads > 0 and not hasattr(dataloader.sampler, "sample_seeded"):
            raise ValueError("sampler_threads > 0 needs a sampler with sample_seeded (FastNeighborSampler)")
        self.device_graph = device_graph
        self.build_blocks = bool(build_blocks)
        self._ring = None
        self._staging = None           # pinned ring for the batches' small arrays (sample_seeded(staging=)): one upload per batch
        self._pos_ring = None          # pinned ring of 16- / 32-bit buffers for the outermost hop's positions
        self._labels_dev = None
        self._memory_bound_set = False
        self.hops = hops
        self.load_stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None   # d_stream
        # In-place native loads of a partly cached store alternate between load_stream and these (DGLL_LOADER_STREAMS, default 2): a
        # batch's uploads and the fetch of its uncached rows use the link while the previous batch's gathers and reduction use HBM
        n_extra = max(0, int(os.environ.get("DGLL_LOADER_STREAMS", "2")) - 1) if self.load_stream is not None else 0
        self._load_streams = [self.load_stream] + [torch.cuda.Stream(self.device) for _ in range(n_extra)]
        self._miss_stages = {}                  # loading stream -> staging buffers of the uncached rows (_bind_miss_stage)
        self._thread = None
        self._error = None
        self._stop = threading.Event()          # set when a stage failed or the consumer left: every producer winds down
        self._static_step = None                # use_static_sets(): batches are written into a captured step's input sets in place
        self._pool = None                       # the native sampler pool of the running epoch (SamplerPool), when it applies
        self._next_set = 1
        self._native_loader = os.environ.get("DGLL_NATIVE_LOADER", "1") != "0"    # in-place loading through ONE native call per batch
        self.load_seconds, self.load_batches = 0.0, 0

    # ---- producer stage 1: sampling (buffer_queues.py:22-46) ---------------------------------------------------
    def _sample(self):
        try:
            it, step = iter(self.dataloader), 0
            while not self._stop.is_set():
                with rng("sample"):
                    item = next(it, None)
                if item is None:
                    break
                inp, outp, subgs = item
                self.sampled.put((step, inp, outp, subgs))            # blocks while the queue is full
                step += 1
        except BaseException as exc:  # noqa: BLE001  (surface producer failures in the consumer)
            self._fail(exc)
        finally:
            self.sampled.put(_DONE)

    def _fail(self, exc):
        """First failure wins; every stage sees the stop flag at its next step."""
        if not isinstance(exc, PipelineStopped) and self._error is None:
            self._error = exc
        self._stop.set()

    def _sample_seeded_worker(self, t, handoff, n_batches):
        """Thread t of K draws batches t, t + K, ... whole, each under its own seed.  Shared between the threads: the sampler's CSR
        copy of the adjacency (read-only; built before the threads start, FastNeighborSampler.prepare), the pinned rings (queues)
        and the stop flag -- a worker that fails sets it, the others leave at their next batch or buffer wait."""
        from .sampling.fast_sampler import batch_seed

        dl = self.dataloader
        try:
            for i in range(t, n_batches, self.sampler_threads):
                if self._stop.is_set() or handoff.closed:
                    handoff.close(0)             # the thread waiting in get() for this worker's next batch must not wait for ever
                    break
                seeds = dl.train_nodes[i * dl.batch_size:(i + 1) * dl.batch_size]
                buf = self._ring.acquire if (self._ring is not None and self._pos_ring is None) else None
                stg = self._staging.acquire if self._staging is not None else None
                with rng("sample"):
                    inp, outp, subgs = dl.sampler.sample_seeded(dl.Dgraph, seeds, batch_seed(self.base_seed, self.epoch, i), max_threads=1,
                                                                last_hop_buffer=buf, staging=stg)
                last = subgs[0]
                if self._pos_ring is not None and getattr(last, "pending_positions", None) is not None and getattr(last, "_finish", None) is not None:
                    # the outermost hop's neighbour POSITIONS (< the maximum degree) leave the host as 16- or 32-bit integers: they are the
                    # largest upload of a batch (2.2 M entries at the Reddit shape: 17.6 MB as int64 next to 40 MB of cache misses on a
                    # link that is the pipeline's bound at a 50 % cache); narrowed here, on the sampling thread, into pinned memory
                    n_pos = int(last._src.shape[0])
                    got = self._pos_ring.acquire(n_pos)
                    if got is not None:
                        small, tok = got
                        np.copyto(small, last._src.numpy(), casting="unsafe")
                        last.positions_compact = (torch.from_numpy(small), tok)
                if not handoff.put(i, (i, inp, outp, subgs)):
                    break
        except BaseException as exc:  # noqa: BLE001
            self._fail(exc)
            handoff.close(0)

    def _pool_applicable(self):
        dl = self.dataloader
        s = getattr(dl, "sampler", None)
        if s is None or self.sampler_threads <= 0:
            return False
        return (self.sampler_threads > 0 and os.environ.get("DGLL_NATIVE_SAMPLER_POOL", "1") != "0" and self.device_graph is not None
                and self.load_stream is not None and self.hops == "sampled" and self.cache is not None and self.build_blocks
                and getattr(s, "defer_last_hop", False) and hasattr(s, "staging_entries") and hasattr(s, "_csr")
                and all(f is not None for f in s.fanouts) and 0 <= self.base_seed < (1 << 24) and len(dl) < (1 << 20)
                and len(s.fanouts) <= 8)

    def _sample_pool(self):
        """Producer stage 1 on the NATIVE sampler pool (SamplerPool): this thread only dequeues finished batches, in order, wraps the
        slot buffers as the views the loading stage expects and hands them on."""
        from .sampling.base_sampler import sugbraph
        from .sampling.fast_sampler import StagedBatch

        pool = self._pool
        L, off = pool.L, pool.offsets
        try:
            while not self._stop.is_set():
                with rng("sample"):
                    got = pool.next()
                if got is None:
                    break
                i, slot, rows, n_src = got
                buf = pool.staged[slot]
                staged = StagedBatch(slot, buf.numpy(), off)
                staged.rows = list(rows)
                subgs = []
                for h in range(L):
                    ptr = buf[off["ptr"][h]:off["ptr"][h] + rows[h] + 1]
                    if h < L - 1:
                        sg = sugbraph(buf[off["src"][h]:off["src"][h] + n_src[h]], None, ptr)
                    else:       # the outermost hop: positions, narrowed, in the slot's second buffer; translated on the device
                        sg = sugbraph(_Positions(n_src[h]), None, ptr, finish=lambda: (_ for _ in ()).throw(
                            RuntimeError("the native sampler pool hands the outermost hop over for a DEVICE-side translation")))
                        hop_seeds = buf[off["seeds"]:off["seeds"] + rows[0]] if L == 1 else buf[off["src"][h - 1]:off["src"][h - 1] + rows[h]]
                        sg.pending_positions = (hop_seeds, None)
                        sg.positions_compact = (pool.pos[slot][:n_src[h]], slot)
                    sg.max_degree = pool.order[h]
                    sg.staged = staged
                    subgs.insert(0, sg)
                seeds = pool.train[i * pool.batch_size:i * pool.batch_size + rows[0]]
                self.sampled.put((i, None, seeds, subgs))
        except BaseException as exc:  # noqa: BLE001
            self._fail(exc)
        finally:
            self.sampled.put(_DONE)

    def _sample_threaded(self):
        """Producer stage 1 in the per-batch-seeded mode: K workers -> OrderedHandoff -> the hand-over queue of the loader."""
        n_batches = len(self.dataloader)
        if hasattr(self.dataloader.sampler, "prepare"):            # the CSR copy of a list-of-lists DGraph: once, before any worker
            self.dataloader.sampler.prepare(self.dataloader.Dgraph)
        handoff = OrderedHandoff(capacity=2 * self.sampler_threads)
        workers = [threading.Thread(target=self._sample_seeded_worker, args=(t, handoff, n_batches), name="dgll-sampler-%d" % t,
                                    daemon=True) for t in range(self.sampler_threads)]
        for w in workers:
            w.start()
        try:
            taken = 0
            while taken < n_batches and not self._stop.is_set():
                item = handoff.get()
                if item is None:
                    break
                self.sampled.put(item)
                taken += 1
        except BaseException as exc:  # noqa: BLE001
            self._fail(exc)
        finally:
            handoff.close(0)                 # put() stops blocking and reports the closure; buffer waits end on the stop flag
            if taken < n_batches:
                self._stop.set()
            self.sampled.put(_DONE)          # before the joins: the loader must not wait for workers that are winding down
            for w in workers:
                w.join()

    def _hop_ids(self, b):
        """hops = "sampled": [seeds, sources around hop 0, ..., outermost sources]; the outermost list comes from the device-side
        translation when the sampler left it as positions and the graph's index arrays are on the device."""
        L = len(b.subgraphs)
        ids = [b.output_nodes] + [b.subgraphs[L - 1 - h].src_nodes() for h in range(L - 1)]
        last = b.subgraphs[0]
        if self.device_graph is not None and self.load_stream is not None and getattr(last, "pending_positions", None) is not None \
                and getattr(last, "_finish", None) is not None:
            ids.append(self._translate_on_device(last))
        else:
            ids.append(last.src_nodes())
            self._drop_compact(last)
            self._late_release = getattr(last, "buffer_token", None)    # a pinned buffer still to be uploaded: freed with b.ready
        return ids

    def _drop_compact(self, sg):
        c = getattr(sg, "positions_compact", None)
        if c is not None:                                   # narrowed positions that no upload will use: back to the ring
            self._pos_ring.release(c[1], None)
            sg.positions_compact = None

    def _translate_on_device(self, sg, device_inputs=None):
        """positions -> neighbour ids on the loading stream: ids[k] = indices[indptr[seed(k)] + position[k]]."""
        indptr, indices = self.device_graph
        pos = sg._src
        compact = getattr(sg, "positions_compact", None)
        with torch.cuda.stream(self.load_stream):
            if device_inputs is not None and len(device_inputs) == 3:
                # the hop's seeds and ROW POINTERS are on the device already (views of the staged upload): ONE launch
                # (dgll_hip_translate_positions) instead of five torch ones; positions leave the host as 16- / 32-bit integers
                seeds_d, ptr_d, n_rows = device_inputs
                src = compact[0] if compact is not None else pos
                pos_d = src.to(self.device, non_blocking=True)
                ids = torch.empty(int(pos.numel()), dtype=torch.int64, device=self.device)
                from . import _lib

                with torch.cuda.device(self.device):
                    _lib.check(_lib.lib.dgll_hip_translate_positions(self.load_stream.cuda_stream, indptr.data_ptr(), indices.data_ptr(),
                                                                     seeds_d.data_ptr(), ptr_d.data_ptr(), int(n_rows), pos_d.data_ptr(),
                                                                     pos_d.element_size(), ids.data_ptr()), "dgll_hip_translate_positions")
            else:
                hop_seeds, counts = sg.pending_positions if device_inputs is None else device_inputs
                seeds_d = hop_seeds.to(self.device, non_blocking=True)
                cnt_d = counts.to(self.device, non_blocking=True)
                if compact is not None:
                    pos_d = compact[0].to(self.device, non_blocking=True).to(torch.int64)
                else:
                    pos_d = pos.to(self.device, non_blocking=True)
                start = indptr[seeds_d]
                ids = indices[torch.repeat_interleave(start, cnt_d, output_size=int(pos.numel())) + pos_d]
            done = torch.cuda.Event()
            done.record(self.load_stream)
        if compact is not None:
            self._pos_ring.release(compact[1], done)
            sg.positions_compact = None
        elif self._ring is not None:
            self._ring.release(getattr(sg, "buffer_token", None), done)
        # the pinned buffer goes back to the ring: from here on the subgraph holds its source ids ON THE DEVICE (src_nodes()) and no
        # per-edge destination list (the structure is in sg.indptr)
        sg._finish, sg._src, sg._dst, sg.pending_positions = None, ids, None, None
        return ids

    # ---- producer stage 2: feature loading --------------------------------------------------------------------
    def _load(self):
        prof = None
        if os.environ.get("DGLL_MB_PROFILE_LOADER"):      # diagnostics: where this thread's host time goes (printed to stderr at its end)
            import cProfile

            prof = cProfile.Profile()
            prof.enable()
        try:
            self._load_loop()
        finally:
            if prof is not None:
                import pstats
                import sys

                prof.disable()
                pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(40)

    def _load_loop(self):
        try:
            while True:
                item = self.sampled.get()
                if item is _DONE:
                    break
                if self._stop.is_set():          # a stage failed or the consumer left: drain without loading
                    self._release_unloaded(item)
                    continue
                t_load = time.perf_counter()
                b = Batch()
                b.step, b.input_nodes, b.output_nodes, b.subgraphs = item
                staged = getattr(b.subgraphs[0], "staged", None) if self.load_stream is not None else None
                if staged is not None:
                    with rng("gpu-load"):
                        self._load_staged(b, staged)
                    self.load_seconds += time.perf_counter() - t_load
                    self.load_batches += 1
                    self.queue.put(b)
                    continue
                self._load_generic(b)
                self.load_seconds += time.perf_counter() - t_load     # this thread's host time per batch (diagnostics)
                self.load_batches += 1
                self.queue.put(b)                                   # blocks while the queue is full
        except BaseException as exc:  # noqa: BLE001
            self._fail(exc)
            while self.sampled.get() is not _DONE:                  # let the sampling thread run to its end (it stops at the flag)
                pass
        finally:
            self.queue.put(_DONE)

    def _load_generic(self, b):
        """The loading stage for a batch whose arrays arrive as host tensors (no staging buffer)."""
        with rng("gpu-load"):
            if self.hops == "sampled":
                id_lists = self._hop_ids(b)
                b.input_nodes = id_lists[-1]
            else:
                if hasattr(b.input_nodes, "resolve"):      # FastNeighborSampler(defer_last_hop=True): the outermost hop's
                    b.input_nodes = b.input_nodes.resolve()    # positions become ids here, off the sampling thread
                id_lists = self.hops(b) if self.hops is not None else [b.input_nodes]
            inp, outp = b.input_nodes, b.output_nodes
            if self.load_stream is not None:
                with torch.cuda.stream(self.load_stream):
                    if self.reduce_last_hop is not None:
                        # outermost hop (subgraphs are outermost first): reduced straight out of the cache, never fetched
                        b.features = self._fetch_many(id_lists[:-1]) + [None]
                        if self.record_access:
                            self.cache.record_access(id_lists[-1], stream=self.load_stream)
                        b.last_hop_reduced = self.cache.aggregate_data(id_lists[-1], b.subgraphs[0].indptr,
                                                                       reduce=self.reduce_last_hop, stream=self.load_stream)
                    else:
                        b.features = self._fetch_many(id_lists)
                    if self.labels is not None:
                        b.labels = self.labels[outp].to(self.device, non_blocking=True)
                    if self.build_blocks:
                        L = len(b.subgraphs)
                        b.blocks = [None if (h == L - 1 and self.reduce_last_hop is not None) else b.subgraphs[L - 1 - h].to_block(self.device)
                                    for h in range(L)]
                    b.ready = torch.cuda.Event()
                    b.ready.record(self.load_stream)
                    if self._ring is not None and getattr(self, "_late_release", None) is not None:
                        self._ring.release(self._late_release, b.ready)
                        self._late_release = None
                if not self._memory_bound_set:          # first loaded batch: how many of these fit in the memory budget
                    self._memory_bound_set = True
                    nbytes = sum(t.numel() * t.element_size() for t in list(b.features) + [b.last_hop_reduced] if t is not None)
                    free, _total = torch.cuda.mem_get_info(self.device)
                    self.queue.set_memory_bound(nbytes, self.memory_fraction * free)
            else:
                b.features = self._fetch_many(id_lists)
                if self.labels is not None:
                    b.labels = self.labels[outp]

    def _release_unloaded(self, item):
        """Pinned buffers of a sampled batch that will not be loaded go back to their rings."""
        try:
            subgs = item[3]
            last = subgs[0]
            staged = getattr(last, "staged", None)
            if staged is not None and self._staging is not None:
                self._staging.release(staged.token, None)
            c = getattr(last, "positions_compact", None)
            if c is not None and self._pos_ring is not None:
                self._pos_ring.release(c[1], None)
            if self._ring is not None:
                self._ring.release(getattr(last, "buffer_token", None), None)
        except Exception:  # noqa: BLE001  (best effort while shutting down)
            pass

    def _load_staged(self, b, staged):
        """The loading stage for a batch whose small arrays arrive packed in one pinned buffer (FastNeighborSampler.sample_seeded(
        staging=)): ONE upload, everything else is views of it on the device -- the seeds and the hops' source ids (fetched with one
        gather), the row pointers (CSR blocks, the outermost hop's fused reduction), the outermost hop's seeds and counts (device-side
        translation of its positions).  The host arrays of the sugbraphs are replaced by these device views: the pinned buffer goes
        back to its ring as soon as the upload has run."""
        L = len(b.subgraphs)
        step = self._static_step
        if step is not None and self._load_into_set(b, staged, step):
            return
        off, n = staged.offsets, staged.rows                       # n[h]: rows of hop h; rows of hop h + 1 = edges of hop h
        with torch.cuda.stream(self.load_stream):
            dev = staged.tensor().to(self.device, non_blocking=True)
            uploaded = torch.cuda.Event()
            uploaded.record(self.load_stream)
            ids = [dev[off["seeds"]:off["seeds"] + n[0]]]
            for h in range(L - 1):
                sg = b.subgraphs[L - 1 - h]
                n_src = int(sg._src.shape[0])
                ids.append(dev[off["src"][h]:off["src"][h] + n_src])
                sg._src = ids[-1]                                      # the host copy lived in the pinned buffer
            ptrs = [dev[off["ptr"][h]:off["ptr"][h] + n[h] + 1] for h in range(L)]
            for h in range(L):
                b.subgraphs[L - 1 - h].indptr = ptrs[h]
                b.subgraphs[L - 1 - h].staged = None
            last = b.subgraphs[0]
            if self.device_graph is not None and getattr(last, "pending_positions", None) is not None and getattr(last, "_finish", None) is not None:
                ids.append(self._translate_on_device(last, device_inputs=(ids[L - 1], ptrs[L - 1], n[L - 1])))
            else:
                ids.append(last.src_nodes())
                self._drop_compact(last)
                self._late_release = getattr(last, "buffer_token", None)
            b.input_nodes = ids[-1]
            self._staging.release(staged.token, uploaded)      # (after the host-side translation above, which reads the hop's seeds)
            if self.reduce_last_hop is not None:
                rows = self._fetch(torch.cat(ids[:-1]))
                b.features = list(rows.split([int(t.numel()) for t in ids[:-1]])) + [None]
                if self.record_access:
                    self.cache.record_access(ids[-1], stream=self.load_stream)
                b.last_hop_reduced = self.cache.aggregate_data(ids[-1], ptrs[L - 1], reduce=self.reduce_last_hop, stream=self.load_stream)
            else:
                dev_ids = [torch.as_tensor(t).reshape(-1).to(self.device, dtype=torch.int64, non_blocking=True) for t in ids]
                rows = self._fetch(torch.cat(dev_ids))
                b.features = list(rows.split([int(t.numel()) for t in dev_ids]))
            if self._labels_dev is not None:
                b.labels = self._labels_dev[ids[0]]
            elif self.labels is not None:
                b.labels = self.labels[b.output_nodes].to(self.device, non_blocking=True)
            b.blocks = [None if (h == L - 1 and self.reduce_last_hop is not None) else b.subgraphs[L - 1 - h].to_block(self.device)
                        for h in range(L)]
            b.ready = torch.cuda.Event()
            b.ready.record(self.load_stream)
            if self._ring is not None and getattr(self, "_late_release", None) is not None:
                self._ring.release(self._late_release, b.ready)
                self._late_release = None
        if not self._memory_bound_set:
            self._memory_bound_set = True
            nbytes = sum(t.numel() * t.element_size() for t in list(b.features) + [b.last_hop_reduced] if t is not None)
            free, _total = torch.cuda.mem_get_info(self.device)
            self.queue.set_memory_bound(nbytes, self.memory_fraction * free)

    def use_static_sets(self, step):
        """From the next loaded batch on, write every batch that fits IN PLACE into one of `step`'s (graphs.GraphedSampledStep, n_sets
        >= 2) input sets 1 .. n_sets-1, round robin: the hop features by one cache gather per hop straight into the set's rows, the
        outermost hop's reduction into its tail, row pointers and labels into its buffers.  The consumer then replays that set's graph
        without copying a byte (Batch.static_set); a set is rewritten only behind the `free` event of the replay that read it.  Needs
        the staged loading path (sampler_threads > 0, hops="sampled", a cache, build_blocks, reduce_last_hop).  None switches it off."""
        if step is not None:
            if len(step.sets) < 2:
                raise ValueError("use_static_sets needs a GraphedSampledStep with n_sets >= 2 (set 0 stays the copy path's)")
            # a set must not come round again before its batch was consumed: the loader runs at most (queue bound + 1) batches ahead
            # of the batch the consumer holds
            bound = self.queue.max_size if self.queue.adaptive else self.queue.size
            if len(step.sets) - 1 < bound + 2:
                raise ValueError("use_static_sets: %d in-place sets for a loaded-batch queue of up to %d: need at least %d (n_sets = %d)" % (
                    len(step.sets) - 1, bound, bound + 2, bound + 3))
        self._next_set = 1
        self._static_step = step

    def _load_into_set(self, b, staged, step):
        """_load_staged's work with the outputs landing in a static input set of the captured step.  False: the batch does not fit."""
        L = len(b.subgraphs)
        off, n = staged.offsets, staged.rows
        n_src = [int(b.subgraphs[L - 1 - h]._src.shape[0]) for h in range(L)]          # edges of hop h = rows of hop h + 1
        if L != step.L or any(n[h] > step.rows[h] for h in range(L)) or any(n_src[h] > step.rows[h + 1] for h in range(L - 1)):
            return False
        last = b.subgraphs[0]
        if self.reduce_last_hop is None or self.device_graph is None or getattr(last, "pending_positions", None) is None \
                or getattr(last, "_finish", None) is None:
            return False
        k = self._next_set
        self._next_set = k + 1 if k + 1 < len(step.sets) else 1
        st = step.sets[k]
        if self._native_loader and self._labels_dev is not None and not self.record_access:
            self._load_into_set_native(b, staged, step, st, k, n, n_src, off)
            return True
        from .graphs import PaddedBlock

        with torch.cuda.stream(self.load_stream):
            if st.free is not None:
                self.load_stream.wait_event(st.free)            # the replay that read this set's previous batch
            dev = staged.tensor().to(self.device, non_blocking=True)
            uploaded = torch.cuda.Event()
            uploaded.record(self.load_stream)
            ids = [dev[off["seeds"]:off["seeds"] + n[0]]]
            for h in range(L - 1):
                sg = b.subgraphs[L - 1 - h]
                ids.append(dev[off["src"][h]:off["src"][h] + n_src[h]])
                sg._src = ids[-1]
            ptrs = [dev[off["ptr"][h]:off["ptr"][h] + n[h] + 1] for h in range(L)]
            for h in range(L):
                b.subgraphs[L - 1 - h].indptr = ptrs[h]
                b.subgraphs[L - 1 - h].staged = None
            ids.append(self._translate_on_device(last, device_inputs=(ids[L - 1], ptrs[L - 1], n[L - 1])))
            b.input_nodes = ids[-1]
            self._staging.release(staged.token, uploaded)
            if self.record_access:
                for t in ids:
                    self.cache.record_access(t, stream=self.load_stream)
            # one launch per hop straight into the set's rows of that hop (one bookkeeping pass for the three)
            self.cache.fetch_data_into(ids[:L], [st.features[h][:n[h]] for h in range(L)], stream=self.load_stream)
            self.cache.aggregate_data(ids[-1], ptrs[L - 1], reduce=self.reduce_last_hop, stream=self.load_stream, out=st.reduced[:n[L - 1]])
            for h in range(L - 1):
                # the set's row pointers in ONE launch: the batch's n[h] + 1 entries, then the edge count for the rows it does not use
                # (the staged block holds rows_cap + 1 >= cap + 1 entries; what lies behind the batch's own is never selected)
                cap = step.rows[h]
                blk = st.blocks[h]
                src = dev[off["ptr"][h]:off["ptr"][h] + cap + 1]
                torch.where(self._arange(cap + 1) <= n[h], src, src[n[h]:n[h] + 1], out=blk.rowptr)
            if n[0] < step.rows[0]:
                st.labels.fill_(-100)
            if self._labels_dev is not None:
                torch.index_select(self._labels_dev, 0, ids[0], out=st.labels[:n[0]])
            elif self.labels is not None:
                st.labels[:n[0]].copy_(self.labels[b.output_nodes].to(self.device, non_blocking=True))
            b.static_set = k
            b.features = [st.features[h][:n[h]] for h in range(L)] + [None]
            b.last_hop_reduced = st.reduced[:n[L - 1]]
            b.labels = st.labels[:n[0]]
            b.blocks = list(st.blocks)
            b.ready = torch.cuda.Event()
            b.ready.record(self.load_stream)
        return True

    def _load_into_set_native(self, b, staged, step, st, k, n, n_src, off):
        """_load_into_set as ONE native call (dgll_hip_load_sampled_batch): both uploads, the positions -> ids translation, the
        hop gathers, the outermost hop's reduction, the padded row pointers and the labels are enqueued on the loading stream from
        C++; what stays in Python is the bookkeeping (rings, events, the views a consumer may look at)."""
        import ctypes as C

        from . import _lib
        from .cache import _dtype_code

        L = len(b.subgraphs)
        last = b.subgraphs[0]
        compact = getattr(last, "positions_compact", None)
        pos_host = compact[0] if compact is not None else last._src
        n_outer = int(last._src.shape[0])
        scratch = getattr(st, "_native", None)
        if scratch is None:                      # per-set device scratch, sized once at the upper bounds
            cap_outer = self.dataloader.batch_size
            for f in self.dataloader.sampler.fanouts:
                cap_outer *= int(f)
            with torch.cuda.stream(self.load_stream):
                scratch = st._native = (torch.empty(int(staged.buffer.shape[0]), dtype=torch.int64, device=self.device),
                                        torch.empty(cap_outer * 8, dtype=torch.uint8, device=self.device),
                                        torch.empty(cap_outer, dtype=torch.int64, device=self.device), _lib.BatchLoad())
        staged_dev, pos_dev, ids_dev, d = scratch
        if int(staged.buffer.shape[0]) > staged_dev.numel() or n_outer > ids_dev.numel():
            raise RuntimeError("a batch larger than the loading stage's scratch (the sampler's fan-outs changed?)")
        # two alternating streams pay when a batch's loads wait on the link (a partly cached store: misses staged over PCIe); with
        # everything cached one stream is as fast or faster (850 against 833 batches/s)
        which = k % len(self._load_streams) if (STAGE_MISSES and not self.cache.full_cached) else 0
        stream = self._load_streams[which]
        if st.free is not None:
            stream.wait_event(st.free)                 # the replay that read this set's previous batch
        cptr, ldc, hptr, ldh, sptr, mptr, counter, keep = self.cache.native_load_begin(stream)
        d.staged_host, d.staged_entries, d.staged_dev = staged.buffer.ctypes.data, int(staged.buffer.shape[0]), staged_dev.data_ptr()
        d.pos_host, d.n_outer, d.pos_bytes, d.pos_dev = pos_host.data_ptr(), n_outer, pos_host.element_size(), pos_dev.data_ptr()
        d.indptr, d.indices = self.device_graph[0].data_ptr(), self.device_graph[1].data_ptr()
        d.n_hops, d.seeds_off = L, off["seeds"]
        for h in range(L):
            d.rows[h] = n[h]
            d.ptr_off[h] = off["ptr"][h]
            d.feat_out[h] = st.features[h].data_ptr()
            if h < L - 1:
                d.src_off[h] = off["src"][h]
                d.rowptr_out[h] = st.blocks[h].rowptr.data_ptr()
                d.rowptr_cap[h] = step.rows[h]
        d.cache, d.ldc, d.host, d.ldh, d.slot, d.host_map = cptr, ldc, hptr, ldh, sptr, mptr
        d.feat, d.dtype = self.cache.total_dim, _dtype_code(st.feat_all)
        d.miss_count = counter.data_ptr() if counter is not None else None
        d.ld_feat = st.feat_all.stride(0)
        d.reduced_out, d.ld_reduced = st.reduced.data_ptr(), st.reduced.stride(0)
        d.reduce = _lib.REDUCE_MEAN if self.reduce_last_hop == "mean" else _lib.REDUCE_SUM
        d.ids_out = ids_dev.data_ptr()
        self._bind_miss_stage(d, stream, which, sptr is not None, n_outer)
        d.labels, d.labels_out, d.labels_cap, d.label_fill = self._labels_dev.data_ptr(), st.labels.data_ptr(), step.rows[0], -100
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib.dgll_hip_load_sampled_batch(stream.cuda_stream, C.byref(d)), "dgll_hip_load_sampled_batch")
        done = torch.cuda.Event()
        done.record(stream)
        self.cache.native_load_end(stream, counter, keep, sum(n[:L]) + n_outer, done)
        self._staging.release(staged.token, done)
        if compact is not None:
            self._pos_ring.release(compact[1], done)
            last.positions_compact = None
        elif self._ring is not None:
            self._ring.release(getattr(last, "buffer_token", None), done)
        # the views a consumer (or a test) may look at: ids and row pointers on the device, as the Python path leaves them
        ids = [staged_dev[off["seeds"]:off["seeds"] + n[0]]]
        for h in range(L - 1):
            ids.append(staged_dev[off["src"][h]:off["src"][h] + n_src[h]])
            b.subgraphs[L - 1 - h]._src = ids[-1]
        for h in range(L):
            sg = b.subgraphs[L - 1 - h]
            sg.indptr = staged_dev[off["ptr"][h]:off["ptr"][h] + n[h] + 1]
            sg.staged = None
        outer = ids_dev[:n_outer]
        last._finish, last._src, last._dst, last.pending_positions = None, outer, None, None
        b.input_nodes = outer
        b.static_set = k
        b.features = [st.features[h][:n[h]] for h in range(L)] + [None]
        b.last_hop_reduced = st.reduced[:n[L - 1]]
        b.labels = st.labels[:n[0]]
        b.blocks = list(st.blocks)
        b.ready = done

    def _bind_miss_stage(self, d, stream, which, partly_cached, n_outer):
        """The staging buffers of `dgll_batch_load` (include/dgll_hip.h) for the loading stream `which`: the outermost hop's uncached rows
        are fetched into HBM ahead of its reduction.  Off with everything (or nothing) cached, or DGLL_LOADER_STAGE_MISSES=0."""
        d.upload_blocks = 0
        if not (STAGE_MISSES and partly_cached and n_outer > 0) or self.cache.full_cached:
            d.stage_map = None
            d.upload_blocks = UPLOAD_BLOCKS_ALONE        # the uploads are the link's only users: a wider grid finishes them sooner
            return
        stages = self._miss_stages
        sg = stages.get(which)
        uncached = max(int(self.cache.node_num) - int(self.cache.cached_num), 1)
        if sg is None or sg["cap"] < min(uncached, sg["cap_outer"], STAGE_CAP if STAGE_CAP > 0 else uncached):
            cap_outer, rows = 0, self.dataloader.batch_size
            for f in [1] + [int(f) for f in reversed(self.dataloader.sampler.fanouts)]:
                rows *= f
                cap_outer += rows                          # ids of all hops of a batch (seeds + every hop's sources)
            cap = min(uncached, cap_outer)                 # every distinct uncached node at most once
            if STAGE_CAP > 0:
                cap = min(cap, STAGE_CAP)                  # nodes past it stay zero-copy reads of the reduction
            feats = self.cache.features
            with torch.cuda.stream(stream):
                sg = stages[which] = {"cap": cap, "cap_outer": cap_outer, "serial": 0,
                                      "map": torch.zeros(int(self.cache.node_num), dtype=torch.int64, device=self.device),
                                      "rows": torch.empty((cap, feats.stride(0)), dtype=feats.dtype, device=self.device),
                                      "list": torch.empty(cap, dtype=torch.int64, device=self.device),
                                      "count": torch.zeros(2, dtype=torch.int32, device=self.device)}
        sg["serial"] = sg["serial"] % 0xFFFFFFF0 + 1       # never 0; a wrap after 4e9 batches meets entries of ancient batches only
        d.stage_map, d.stage_rows, d.ld_stage = sg["map"].data_ptr(), sg["rows"].data_ptr(), sg["rows"].stride(0)
        d.stage_cap, d.stage_list, d.stage_count = sg["cap"], sg["list"].data_ptr(), sg["count"].data_ptr()
        d.stage_serial, d.stage_blocks = sg["serial"], STAGE_BLOCKS

    def _arange(self, m):
        ar = getattr(self, "_ar_cache", None)
        if ar is None or ar.numel() < m:
            ar = self._ar_cache = torch.arange(max(m, 1 << 17), dtype=torch.int64, device=self.device)
        return ar[:m]

    def _fetch_many(self, id_lists):
        """Features of several id lists with ONE gather: the lists' rows are consecutive slices of one buffer (one id upload, one
        launch; GraphSage.forward_sampled stacks the hops of a layer without copying them)."""
        if len(id_lists) <= 1 or self.cache is None:
            # without a cache the rows come from Dgraph.get_features -- an indexing of a HOST tensor, which takes host ids: one
            # fetch per list (joining them on the device, as below, would index a host tensor with device ids)
            return [self._fetch(ids) for ids in id_lists]
        sizes = [int(ids.numel()) for ids in id_lists]
        # the lists are joined ON THE DEVICE: a multi-threaded host torch.cat in this thread wakes the intra-op pool, whose spinning
        # workers took the cores from the sampling thread (its draw went from 3 to 12 ms per batch)
        dev_ids = [torch.as_tensor(ids).reshape(-1).to(self.device, dtype=torch.int64, non_blocking=True) for ids in id_lists]
        rows = self._fetch(torch.cat(dev_ids))
        return list(rows.split(sizes))

    def _fetch(self, ids):
        if self.cache is not None:
            if self.record_access:
                self.cache.record_access(ids, stream=self.load_stream)
            return self.cache.fetch_data(ids, stream=self.load_stream)
        feats = self.dataloader.Dgraph.get_features(ids.cpu() if isinstance(ids, torch.Tensor) and ids.is_cuda else ids)
        return feats.to(self.device, non_blocking=True)

    # ---- consumer side -----------------------------------------------------------------------------------------
    def __iter__(self):
        self._error = None
        self._stop.clear()
        self._pool = None
        if self._pool_applicable():
            dl = self.dataloader
            indptr, indices = dl.sampler._csr(dl.Dgraph)
            max_deg = int((self.device_graph[0][1:] - self.device_graph[0][:-1]).max())
            n_thr = self.sampler_threads
            self._pool = SamplerPool(indptr, indices, dl.train_nodes, dl.batch_size, dl.sampler.fanouts, self.base_seed, self.epoch, n_thr, max_deg)
            self._staging, self._pos_ring, self._ring = _PoolPart(self._pool), _PoolPart(self._pool), None
            if self.labels is not None and self._labels_dev is None:
                self._labels_dev = self.labels.to(self.device)
        elif self.sampler_threads > 0 and self.device_graph is not None and self.load_stream is not None and self._ring is None \
                and self._pos_ring is None \
                and getattr(self.dataloader.sampler, "defer_last_hop", False):
            cap = self.dataloader.batch_size
            for f in self.dataloader.sampler.fanouts:
                cap *= int(f)
            n_buf = 3 * self.sampler_threads + 4                              # more buffers than batches can be in flight before the upload
            max_deg = int((self.device_graph[0][1:] - self.device_graph[0][:-1]).max())
            if max_deg < (1 << 31):
                self._pos_ring = PinnedRing(n_buf, cap, dtype=torch.int16 if max_deg < (1 << 15) else torch.int32, stop=self._stop)
            else:
                self._ring = PinnedRing(n_buf, cap, stop=self._stop)
            if self.hops == "sampled" and self.cache is not None and self.build_blocks and hasattr(self.dataloader.sampler, "staging_entries"):
                self._staging = PinnedRing(3 * self.sampler_threads + 4,
                                           self.dataloader.sampler.staging_entries(self.dataloader.batch_size, self.dataloader.sampler.fanouts),
                                           stop=self._stop)
                if self.labels is not None and self._labels_dev is None:
                    self._labels_dev = self.labels.to(self.device)
        # Three Python threads share the interpreter lock (producer hand-over, loader, consumer) and each drops it around its native
        # calls; a thread coming back from one waits until the holder next yields -- by default only every 5 ms.  A shorter switch
        # interval bounds that wait (DGLL_PIPELINE_SWITCH_INTERVAL seconds, 0 = leave the interpreter's setting alone); restored at the end.
        import sys

        self._old_switch = None
        want = float(os.environ.get("DGLL_PIPELINE_SWITCH_INTERVAL", "0.0002"))
        if want > 0 and want < sys.getswitchinterval():
            self._old_switch = sys.getswitchinterval()
            sys.setswitchinterval(want)
        self._thread = threading.Thread(target=self._sample_pool if self._pool is not None else
                                        (self._sample_threaded if self.sampler_threads > 0 else self._sample),
                                        name="dgll-sample-producer", daemon=True)
        self._loader = threading.Thread(target=self._load, name="dgll-feature-loader", daemon=True)
        self._thread.start()
        self._loader.start()
        finished = False
        try:
            yield from self._consume()
            finished = True
        finally:
            if not finished:                 # the consumer left early (break / exception in its loop): wind the producers down
                self._stop.set()
                while self.queue.get() is not _DONE:
                    pass
            self._thread.join()
            self._loader.join()
            if self._old_switch is not None:
                sys.setswitchinterval(self._old_switch)
                self._old_switch = None
            if self._pool is not None:          # stop + join the native workers; the rings of the next epoch are built afresh
                self.pool_stats = {"threads": self._pool.n_threads, "sample_ms_per_batch": self._pool.sample_ms / max(self._pool.delivered, 1),
                                   "batches": self._pool.delivered}
                self._pool.close()
                self._pool, self._staging, self._pos_ring = None, None, None
        if self._error is not None:
            raise self._error

    def _consume(self):
        while True:
            b = self.queue.get()
            if b is _DONE:
                break
            if b.ready is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(b.ready)                      # compute stream waits for the load stream
                # The tensors were allocated on the load stream and are consumed on `cur`: tell the caching allocator, or
                # the block returns to the load stream's pool when the consumer drops the batch -- while forward/backward
                # kernels reading it may still be queued -- and the loader's next gather could be written into it.
                if b.static_set is None:                     # (a static input set is long-lived: nothing to tell the allocator)
                    for t in list(b.features or ()) + [b.last_hop_reduced]:
                        if t is not None:
                            t.record_stream(cur)
                    for blk in (b.blocks or ()):
                        if blk is not None:
                            blk.rowptr.record_stream(cur)
                            blk.col.record_stream(cur)
                    if b.labels is not None and b.labels.is_cuda:
                        b.labels.record_stream(cur)
            yield b
